"""diagnostic (round 5, call 15): which of batch / vocabulary / history makes the engine-driven world-2 decode step return NaN"""
import os, sys, json
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from zhilight_amd import _lib, build
_lib.lib()
sys.path.insert(0, os.path.dirname(build.refcompile_target()))
import zl_reflinear as ref
from zhilight_amd.llama import ModelConfig
from test_gpu_model import _hf_state
from test_gpu_refcompile import _reference_names_state

def run(batch, vocab, warm, steps=2, layers=2):
    rng = np.random.default_rng(11)
    cfg = ModelConfig(num_layers=layers, dim_model=1024, num_heads=8, dim_head=128, dim_ff=2048, vocab_size=vocab, num_kv_heads=2, eps=1e-5, rope_theta=5e5)
    sd = _hf_state(rng, cfg, 128)
    ref.weight_cache_clear()
    m = ref.RefEngineLLaMA(cfg.num_layers, cfg.dim_model, cfg.num_heads, cfg.num_kv_heads, cfg.dim_head, cfg.dim_ff, cfg.vocab_size, eps=cfg.eps,
                           rope_theta=cfg.rope_theta, quant_type=5, group_size=128, devices=[0, 0])
    m.load(_reference_names_state(sd), "m")
    len_buf = 64
    empty = np.zeros((cfg.num_layers, 0, cfg.num_kv_heads, cfg.dim_head), np.float16)
    for b in range(batch):
        m.set_history(b, len_buf, empty, empty)
    tokens = rng.integers(0, vocab, batch).astype(np.int32)
    out = {"batch": batch, "vocab": vocab, "warm": warm, "layers": layers, "steps": []}
    seq = ([0] * warm) + list(range(steps))
    for step in seq:
        pos = np.full(batch, step, np.int32)
        mask = np.concatenate([(np.arange(len_buf) <= step).astype(np.int8) for _ in range(batch)])
        both = m.decode_step(tokens, pos, mask)
        out["steps"].append({"pos": step, "finite": [float(np.isfinite(both[r]).mean()) for r in range(2)],
                             "equal": bool(np.array_equal(both[0].view(np.uint16), both[1].view(np.uint16))), "errs": m.exchange_errors()})
    del m
    print(json.dumps(out), flush=True)

for args in ((1, 512, 1), (1, 1000, 1), (3, 512, 1), (2, 512, 1), (3, 1000, 1), (3, 512, 0), (4, 512, 1), (3, 512, 1, 2, 1)):
    try:
        run(*args)
    except Exception as e:
        print("FAILED", args, repr(e)[:500], flush=True)
