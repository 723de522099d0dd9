"""round 5: the reference's own zhilight.C (zhilight_amd/_ref/C*.so: src/py_export + src/generator/batch_generator.cpp compiled unmodified on
libzhilight_amd_host.so) driven the way zhilight/dynamic_batch.py drives it -- engine, model, load, BatchGenerator thread, one greedy task."""
import faulthandler, json, os, sys, threading, time
import numpy as np
faulthandler.enable()
faulthandler.dump_traceback_later(150, exit=True)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from zhilight_amd import _lib, build
_lib.lib()
sys.path.insert(0, os.path.dirname(build.binding_target()))
import C
from zhilight_amd.llama import ModelConfig
from test_gpu_model import OracleModel, _hf_state
from test_gpu_refcompile import _reference_names_state
import oracle.zl_oracle as oracle
oracle.lib()

rng = np.random.default_rng(21)
cfg = ModelConfig(num_layers=2, dim_model=1024, num_heads=8, dim_head=128, dim_ff=2048, vocab_size=512, num_kv_heads=2, eps=1e-5, rope_theta=5e5)
g = 128
sd = _hf_state(rng, cfg, g)
state = {k.replace("m.", "llama.", 1): v for k, v in _reference_names_state(sd).items()}
mc = C.ModelConfig({"model_type": "llama", "num_layers": cfg.num_layers, "dim_model": cfg.dim_model, "num_heads": cfg.num_heads, "dim_head": cfg.dim_head,
                    "dim_ff": cfg.dim_ff, "vocab_size": cfg.vocab_size, "eps": cfg.eps, "num_kv_heads": cfg.num_kv_heads, "dtype": "half", "rope_theta": cfg.rope_theta})
qc = C.QuantConfig(5, True, False, g, False)
dist = C.DistConfig(1, "", 1, 0)
engine = C.Engine(0, 8 << 30, dist)
model = C.LLaMA(engine, mc, qc, dist)
model.load_state_dict(state)
print("loaded", flush=True)
dc = C.DynBatchConfig()
dc.max_batch, dc.max_beam_size, dc.task_queue_size, dc.max_total_token = 4, 1, 8, 1024
dc.eos_id, dc.bos_id, dc.unk_id = 1, 2, 0
dc.rag_buffer, dc.flash_attention, dc.ignore_eos = True, True, True
gen = C.BatchGenerator(dc, model)
err = []
def run():
    try:
        gen.run()
    except Exception as e:
        err.append(repr(e))
        print("generator thread:", repr(e)[:2000], flush=True)
th = threading.Thread(target=run, daemon=True)
th.start()
prompt = [int(t) for t in rng.integers(3, cfg.vocab_size, 17)]
n_new = 6
task = C.SearchTask(prompt, 1, n_new, 0.0, 1.0, 1.0, False, 0, 1.0, 1, 1.0, 0, False, 0, 0, 0)
assert gen.submit(task, True)
res = None
t0 = time.time()
while time.time() - t0 < 60 and not err:
    if task.has_result():
        res = task.get_result(1.0)
        break
    time.sleep(0.05)
print("result", res, flush=True)
out = {"errors": err, "result": None}
if res is not None and res[3]:
    tokens = list(res[3][0][0])
    # the oracle's greedy continuation of the same prompt (bos / eos masked at the first step, as apply_repetition_penalty does)
    om = OracleModel(oracle, cfg, sd, g, 1, 64)
    om.rope_kind = "plain"
    logits = om.prefill(0, np.array(prompt, np.int32))
    want = []
    for step in range(n_new):
        row = logits[0].copy()
        if step == 0:
            row[dc.bos_id] = row[dc.eos_id] = -50000
        tok = int(np.argmax(row))
        want.append(tok)
        if step + 1 < n_new:
            logits, _ = om.step(np.array([tok], np.int32), [len(prompt) + step])
    out.update(result=tokens, oracle=want, agree=tokens[-n_new:] == want)
    print("tokens", tokens, "oracle", want, flush=True)
gen.stop()
th.join(timeout=10)
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(root, "gpurun_out", "binding_try.json"), "w"), indent=1)
print(json.dumps(out), flush=True)
os._exit(0 if out.get("agree") else 1)
