"""Child process of tests/test_gpu_binding.py: the reference's own `zhilight.C` extension module (zhilight_amd/_ref/C*.so = src/py_export/*.cpp +
src/generator/batch_generator.cpp compiled unmodified, linked on libzhilight_amd_host.so) driven the way zhilight/dynamic_batch.py drives it:
Engine -> LLaMA -> load_state_dict -> BatchGenerator on its own thread -> SearchTask submit / batch_search.  Prints ONE JSON line.
(The module travels prebuilt; nothing here reads /root/reference.)"""
import faulthandler
import json
import os
import sys
import threading
import time

import numpy as np

faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("ZL_BINDING_WATCHDOG", "90")), exit=True)   # a scheduler thread that never answers must not hang the box
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from zhilight_amd import _lib, build          # noqa: E402

_lib.lib()
sys.path.insert(0, os.path.dirname(build.binding_target()))
import C                                    # noqa: E402
import oracle.zl_oracle as oracle           # noqa: E402
from test_gpu_model import OracleModel, _hf_state          # noqa: E402
from test_gpu_refcompile import _reference_names_state    # noqa: E402
from zhilight_amd.llama import ModelConfig  # noqa: E402

oracle.lib()
BOS, EOS = 2, 1


def greedy_oracle(cfg, sd, g, prompt, n_new):
    """the CPU oracle's greedy continuation; bos / eos are pushed to -50000 at the first step, as SearcherImplV1::apply_repetition_penalty does
    (src/generator/batch_generator.cpp:1639-1660)"""
    om = OracleModel(oracle, cfg, sd, g, 1, 128)
    om.rope_kind = "plain"
    logits = om.prefill(0, np.array(prompt, np.int32))
    out, margins = [], []
    for step in range(n_new):
        row = logits[0].astype(np.float64).copy()
        if step == 0:
            row[BOS] = row[EOS] = -50000
        order = np.argsort(-row, kind="stable")
        out.append(int(order[0]))
        margins.append(float(row[order[0]] - row[order[1]]) / float(np.abs(row).max()))
        if step + 1 < n_new:
            logits, _ = om.step(np.array([out[-1]], np.int32), [len(prompt) + step])
    return out, margins


def fixture_main(case):
    """ZL_BINDING_FIXTURE=<case>: drive `C` with EXACTLY what the reference's Python layer produced for a synthetic HF checkpoint --
    tests/golden/python_layer_<case>.{json,npz}, written by tools/gen_python_layer_fixture.py where /root/reference exists: the dict
    zhilight/llama.py hands to C.ModelConfig, quant_config_to_c's arguments, the environment switches zhilight/quant.py / llama.py set
    (desc_act: GPTQ_KERNEL_ALGO=0 -> nn::gptq::gptq_gemm), the renamed / re-viewed state dict of
    LLaMALoader + load_state_dict_pt, DynamicBatchConfig.c_config()'s fields, to_c_task's SearchTask arguments, and the engine /
    dist arguments LLaMA.__init__ uses by default (device -1, memory limit 0, DistConfig())."""
    from tools.gen_python_layer_fixture import hf_tensors, paths
    jp, npz = paths(case)
    meta = json.load(open(jp))
    out = {"errors": [], "case": case}
    for k, v in meta["env"].items():
        os.environ[k] = v
    config = dict(meta["config"])
    if meta.get("quantization_config"):
        config["quantization_config"] = meta["quantization_config"]
    with np.load(npz) as z:
        state = {k: z[k] for k in z.files}
    cfg, sd, _ = hf_tensors(case)
    g = 128
    dense = cfg.dtype == "bfloat16"                     # the MiniCPM case: unquantised bf16, tied lm_head
    # the fixture IS the synthetic checkpoint under the reference's names: tie it to the tensors the oracle model is built from
    named = dict(sd, **{"lm_head.weight": sd["model.embed_tokens.weight"]}) if dense else sd
    mine = {k.replace("m.", "llama.", 1): v for k, v in _reference_names_state(named).items()}
    if dense:
        del mine["llama.lm_head.weight"]
    out["names_match"] = sorted(mine) == sorted(state)
    out["tensors_match"] = out["names_match"] and all(np.array_equal(np.asarray(mine[k]).view(np.uint8), state[k].view(np.uint8)) for k in mine)
    mc = C.ModelConfig(config)
    dist = C.DistConfig(-1, "", 1, 0)                    # DistConfig().to_c_config() (zhilight/config/dist_config.py)
    engine = C.Engine(-1, int(os.environ.get("ZL_BINDING_MEM", "0")), dist)   # LLaMA.__init__ defaults (zhilight/llama.py:121-141)
    qa = meta["quant_config_to_c"]
    model = C.LLaMA(engine, mc, C.QuantConfig(int(qa[0]), qa[1], bool(qa[2]), int(qa[3]), bool(qa[4])), dist)
    model.load_state_dict(state)
    dc = C.DynBatchConfig()
    for f, v in meta["dyn_batch_config"].items():
        setattr(dc, f, v)
    gen = C.BatchGenerator(dc, model)
    errors = out["errors"]

    def run():
        try:
            gen.run()
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e)[:2000])

    th = threading.Thread(target=run, daemon=True)
    th.start()
    prompt = meta["prompt"]
    t1 = C.SearchTask(prompt, *meta["search_task_args"])
    assert gen.submit(t1, True)
    t0, r1 = time.time(), None
    while time.time() - t0 < 60 and not errors:
        if t1.has_result():
            r1 = t1.get_result(1.0)
            break
        time.sleep(0.02)
    # the oracle: the CPU model over the same checkpoint; a desc_act checkpoint through its dense W16 matrices (reconstruct_gptq's)
    om_sd = dict(sd)
    w16 = None
    if any(k.endswith(".g_idx") for k in sd):
        w16 = {}
        for key in [k for k in sd if k.endswith(".qweight")]:
            base = key[:-8]
            zp1 = oracle.gptq_increase_zero(sd[base + ".qzeros"].view(np.uint32))
            w_kn = oracle.gptq_reconstruct(sd[key].view(np.uint32), zp1, sd[base + ".scales"].view(np.uint16), sd[base + ".g_idx"])
            w16[base] = np.ascontiguousarray(w_kn.T)
    n_new = meta["search_task_args"][1]
    if dense:
        # the oracle: the CPU dense model (bf16 rounding points, MiniCPM scalings) fed the prompt token by token, then its own picks
        from test_gpu_model import OracleDenseModel
        om = OracleDenseModel(oracle, cfg, sd, 1, 128, 1)
        logits = None
        for i, tok in enumerate(prompt):
            logits = om.step(np.array([tok], np.int32), [i])
        want, margins = [], []
        for step in range(n_new):
            row = np.asarray(logits[0], np.float64).copy()
            if step == 0:
                row[meta["dyn_batch_config"]["bos_id"]] = row[meta["dyn_batch_config"]["eos_id"]] = -50000
            order = np.argsort(-row, kind="stable")
            want.append(int(order[0]))
            margins.append(float(row[order[0]] - row[order[1]]) / float(np.abs(row).max()))
            if step + 1 < n_new:
                logits = om.step(np.array([want[-1]], np.int32), [len(prompt) + step])
        out["greedy"] = {"got": list(r1[3][0][0]) if r1 and r1[3] else None, "oracle": want, "margins": margins,
                         "first_token_delay_ms": r1[3][0][3] if r1 and r1[3] else None}
        out["env"] = meta["env"]
        print("BINDING_RESULT " + json.dumps(out), flush=True)
        gen.stop()
        th.join(timeout=10)
        os._exit(0)
    om = OracleModel(oracle, cfg, om_sd, g, 1, 128)
    om.rope_kind = "plain"
    if w16 is not None:
        om.w16 = w16
    logits = om.prefill(0, np.array(prompt, np.int32))
    want, margins = [], []
    for step in range(n_new):
        row = logits[0].astype(np.float64).copy()
        if step == 0:
            row[meta["dyn_batch_config"]["bos_id"]] = row[meta["dyn_batch_config"]["eos_id"]] = -50000
        order = np.argsort(-row, kind="stable")
        want.append(int(order[0]))
        margins.append(float(row[order[0]] - row[order[1]]) / float(np.abs(row).max()))
        if step + 1 < n_new:
            logits, _ = om.step(np.array([want[-1]], np.int32), [len(prompt) + step])
    out["greedy"] = {"got": list(r1[3][0][0]) if r1 and r1[3] else None, "oracle": want, "margins": margins,
                     "first_token_delay_ms": r1[3][0][3] if r1 and r1[3] else None}
    out["env"] = meta["env"]
    print("BINDING_RESULT " + json.dumps(out), flush=True)
    gen.stop()
    th.join(timeout=10)
    os._exit(0)


def main():
    if os.environ.get("ZL_BINDING_FIXTURE"):
        return fixture_main(os.environ["ZL_BINDING_FIXTURE"])
    rng = np.random.default_rng(21)
    cfg = ModelConfig(num_layers=2, dim_model=1024, num_heads=8, dim_head=128, dim_ff=2048, vocab_size=512, num_kv_heads=2, eps=1e-5, rope_theta=5e5)
    g = 128
    sd = _hf_state(rng, cfg, g)
    state = {k.replace("m.", "llama.", 1): v for k, v in _reference_names_state(sd).items()}
    mc = C.ModelConfig({"model_type": "llama", "num_layers": cfg.num_layers, "dim_model": cfg.dim_model, "num_heads": cfg.num_heads, "dim_head": cfg.dim_head,
                        "dim_ff": cfg.dim_ff, "vocab_size": cfg.vocab_size, "eps": cfg.eps, "num_kv_heads": cfg.num_kv_heads, "dtype": "half",
                        "rope_theta": cfg.rope_theta})
    dist = C.DistConfig(1, "", 1, 0)
    engine = C.Engine(0, 8 << 30, dist)
    model = C.LLaMA(engine, mc, C.QuantConfig(5, True, False, g, False), dist)
    model.load_state_dict(state)
    dc = C.DynBatchConfig()
    dc.max_batch, dc.max_beam_size, dc.task_queue_size, dc.max_total_token = 4, 1, 8, 1024
    dc.eos_id, dc.bos_id, dc.unk_id = EOS, BOS, 0
    dc.rag_buffer, dc.flash_attention, dc.ignore_eos = True, True, True
    gen = C.BatchGenerator(dc, model)
    errors = []

    def run():
        try:
            gen.run()
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e)[:2000])

    th = threading.Thread(target=run, daemon=True)
    th.start()

    def task(prompt, n_new, beam=1, top_p=1.0, top_k=0, seed=0, temperature=1.0):
        # (tokens, beam_size, max_length, presence_penalty, repetition_penalty, ngram_penalty, diverse, seed, temperature, num_results, top_p, top_k,
        #  bee_answer_multi_span, top_logprobs, stream, output_hidden_states): src/py_export/py_batch_generator.cpp:38-56
        return C.SearchTask([int(t) for t in prompt], beam, n_new, 0.0, 1.0, 1.0, False, seed, temperature, 1, top_p, top_k, False, 0, 0, 0)

    def wait(t, seconds=60):
        t0 = time.time()
        while time.time() - t0 < seconds and not errors:
            if t.has_result():
                return t.get_result(1.0)
            time.sleep(0.02)
        return None

    out = {"errors": errors}
    # 1. one greedy task through submit / get_result
    p1 = rng.integers(3, cfg.vocab_size, 17)
    t1 = task(p1, 6)
    assert gen.submit(t1, True)
    r1 = wait(t1)
    want1, m1 = greedy_oracle(cfg, sd, g, p1, 6)
    out["greedy"] = {"got": list(r1[3][0][0]) if r1 and r1[3] else None, "oracle": want1, "margins": m1, "first_token_delay_ms": r1[3][0][3] if r1 and r1[3] else None}
    print("BINDING_RESULT " + json.dumps(out), flush=True)        # (the proven case first: what follows may only add to it)
    # 2. sampling (top_p < 1, temperature): the host-side sampler behind random_sampler_gpu with the counter-based generator -- one task at a
    #    time, which is as far as the reference's own assertion at batch_generator.cpp:751 lets a beam-1 configuration go (fill_last_hidden_states
    #    slices max_beam_size rows of the hidden states and then demands max_batch_active of them: two active tasks trip it in the reference
    #    itself, profiles/r05_zhilight_C_batch_assertion.log)
    if not errors:
        draws = []
        for seed in (1234, 1234, 99):
            ts = task(p1, 6, top_p=0.9, seed=seed, temperature=0.8)
            assert gen.submit(ts, True)
            rs = wait(ts, 20)
            draws.append(list(rs[3][0][0]) if rs and rs[3] else None)
        out["sampling"] = {"draws": draws, "vocab": cfg.vocab_size}
        print("BINDING_RESULT " + json.dumps(out), flush=True)
    if os.environ.get("ZL_BINDING_EXTRA") == "1":
        # 3. three tasks of different lengths submitted back to back: dynamic batching (a prompt joins while the others decode); polled, never a
        #    blocking wait inside the module -- a scheduler thread that died leaves its message in `errors` (today: the :751 assertion)
        prompts = [rng.integers(3, cfg.vocab_size, n) for n in (5, 40, 23)]
        tasks = [task(p, 5) for p in prompts]
        for t in tasks:
            assert gen.submit(t, True)
        out["batch"] = []
        for p, t in zip(prompts, tasks):
            r = wait(t, 20)
            want, m = greedy_oracle(cfg, sd, g, p, 5)
            out["batch"].append({"got": list(r[3][0][0]) if r and r[3] else None, "oracle": want, "margins": m})
        print("BINDING_RESULT " + json.dumps(out), flush=True)
    gen.stop()
    th.join(timeout=10)
    print("BINDING_RESULT " + json.dumps(out), flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
