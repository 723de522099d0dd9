"""Golden fixture of what the reference's PYTHON layer hands to `zhilight.C` for a synthetic HF checkpoint.

The reference's Python sources cannot travel to the GPU box (only /root/reference in this container has them), and this container
has no GPU -- so "zhilight.llama drops in unchanged" is checked in two halves that meet in this fixture:
  * HERE (CPU): the reference's own `zhilight` package -- every .py symlinked into a temporary package directory next to the built
    `C*.so` (zhilight_amd/_ref: the reference's src/py_export + batch_generator.cpp compiled unmodified on the MI355X host library),
    nothing copied, nothing edited -- is imported and run as far as it goes without a device: config adaptation
    (zhilight/config/adapter.py, zhilight/llama.py:59-113 _get_config), quantisation config + the environment switches it sets
    (zhilight/quant.py:27-91), the safetensors loader and its HF -> internal renaming (zhilight/loader.py:250-358), the dtype views of
    LLaMA.load_state_dict_pt (zhilight/llama.py:186-208), DynamicBatchConfig.c_config / DynamicBatchGenerator.to_c_task
    (zhilight/dynamic_batch.py:62-81, 424-447).  What they produce -- the dict for C.ModelConfig, the C.QuantConfig arguments, the
    environment delta, the state dict for C.LLaMA.load_state_dict, the C.DynBatchConfig fields, the C.SearchTask arguments -- is
    written to tests/golden/python_layer_<case>.{json,npz}.
  * on the GPU box: tests/test_gpu_zz_binding.py feeds exactly those values to the same `C*.so` and compares the generated tokens
    with the CPU oracle model built from the same synthetic checkpoint.
tests/test_host_logic.py::test_python_layer_fixture_is_what_the_reference_produces re-runs this file's generation in memory where
/root/reference exists and compares it with the committed fixture.

usage: python tools/gen_python_layer_fixture.py [--check]      (writes / verifies tests/golden/python_layer_*.{json,npz})
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("ZL_REFERENCE", "/root/reference")
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    # BASELINE configs[1] in miniature: Llama-shaped, GPTQ-Int4 g128
    "llama_gptq": dict(seed=5, layers=2, dim=256, heads=2, kv_heads=1, dim_head=128, ff=512, vocab=256, desc_act=False),
    # the same checkpoint format with desc_act: zhilight/quant.py:73-76 switches the GPTQ_KERNEL_ALGO=0 route on
    "llama_gptq_desc_act": dict(seed=6, layers=2, dim=256, heads=2, kv_heads=1, dim_head=128, ff=512, vocab=256, desc_act=True),
    # BASELINE configs[0] in miniature: MiniCPM ("cpm_dragonfly": scale_emb, scale_depth residuals, dim_model_base logit scaling, tied
    # lm_head, 64-wide heads), unquantised bf16 -- the checkpoint in the model family's own config keys, bf16 tensors in the safetensors
    "minicpm_bf16": dict(seed=7, layers=2, dim=256, heads=4, kv_heads=4, dim_head=64, ff=512, vocab=256, desc_act=False, minicpm=True),
}


def reference_package():
    """a temporary `zhilight` package: symlinks to the reference's own files + the built extension module"""
    from zhilight_amd import build
    so = build.binding_target()
    if not os.path.isdir(os.path.join(REFERENCE, "zhilight")) or not os.path.exists(so):
        return None
    top = tempfile.mkdtemp(prefix="zl_refpy_")
    pkg = os.path.join(top, "zhilight")
    os.mkdir(pkg)
    for f in os.listdir(os.path.join(REFERENCE, "zhilight")):
        os.symlink(os.path.join(REFERENCE, "zhilight", f), os.path.join(pkg, f))
    os.symlink(so, os.path.join(pkg, os.path.basename(so)))
    return top


def hf_tensors(case):
    """(ModelConfig, HF-named tensors, HF config.json dict) of a synthetic checkpoint -- needs nothing of the reference"""
    from test_gpu_model import _hf_state
    from zhilight_amd.llama import ModelConfig
    c = CASES[case]
    if c.get("minicpm"):
        from test_gpu_model import _dense_state
        cfg = ModelConfig(num_layers=c["layers"], dim_model=c["dim"], num_heads=c["heads"], dim_head=c["dim_head"], dim_ff=c["ff"],
                          vocab_size=c["vocab"], num_kv_heads=c["kv_heads"], eps=1e-5, rope_theta=1e4, dtype="bfloat16", scale_emb=12.0,
                          scale_depth=1.4, dim_model_base=256, tie_lm_head=True)
        rng = np.random.default_rng(c["seed"])
        sd32 = _dense_state(rng, cfg)
        sd32["model.embed_tokens.weight"] = (sd32["model.embed_tokens.weight"].astype(np.float32) * 0.08).astype(np.float16)
        del sd32["lm_head.weight"]                      # tied: the checkpoint has none
        import oracle.zl_oracle as zo
        sd = {k: zo.f32_to_bf16(v.astype(np.float32)) for k, v in sd32.items()}          # bf16 bit patterns (uint16)
        native = {"model_type": "cpm_dragonfly", "num_layers": c["layers"], "dim_model": c["dim"], "num_heads": c["heads"],
                  "num_kv_heads": c["kv_heads"], "dim_head": c["dim_head"], "dim_ff": c["ff"], "vocab_size": c["vocab"], "eps": 1e-5,
                  "rope_theta": 1e4, "scale_emb": 12.0, "scale_depth": 1.4, "dim_model_base": 256, "_dtype": "bf16",
                  "activate_fn": "silu", "max_token": 2048, "bos_token_id": 2, "eos_token_id": 1}
        return cfg, sd, native
    cfg = ModelConfig(num_layers=c["layers"], dim_model=c["dim"], num_heads=c["heads"], dim_head=c["dim_head"], dim_ff=c["ff"],
                      vocab_size=c["vocab"], num_kv_heads=c["kv_heads"], eps=1e-5, rope_theta=5e5)
    rng = np.random.default_rng(c["seed"])
    sd = _hf_state(rng, cfg, 128)
    if c["desc_act"]:
        # a desc_act checkpoint: every quantised linear carries g_idx; the rows of a group are scattered by a random permutation
        for name in [k[:-8] for k in list(sd) if k.endswith(".qweight")]:
            k_in = sd[name + ".qweight"].shape[0] * 8
            order = rng.permutation(k_in)
            g_idx = np.empty(k_in, np.int32)
            g_idx[order] = np.arange(k_in, dtype=np.int32) // 128
            sd[name + ".g_idx"] = g_idx
    hf_cfg = {"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_size": c["dim"], "intermediate_size": c["ff"],
              "num_attention_heads": c["heads"], "num_key_value_heads": c["kv_heads"], "num_hidden_layers": c["layers"],
              "vocab_size": c["vocab"], "rms_norm_eps": 1e-5, "rope_theta": 5e5, "torch_dtype": "float16", "max_position_embeddings": 2048,
              "bos_token_id": 2, "eos_token_id": 1,
              "quantization_config": {"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": bool(c["desc_act"]), "sym": False}}
    return cfg, sd, hf_cfg


def hf_checkpoint(case, directory):
    """the synthetic checkpoint as an HF-format directory (config.json + model.safetensors)"""
    from safetensors.numpy import save_file
    _, sd, hf_cfg = hf_tensors(case)
    with open(os.path.join(directory, "config.json"), "w") as fh:
        json.dump(hf_cfg, fh)
    if CASES[case].get("minicpm"):                     # bf16 tensors: numpy has no such dtype, the torch writer does
        import torch
        from safetensors.torch import save_file as save_torch
        save_torch({k: torch.from_numpy(np.ascontiguousarray(v).view(np.int16)).view(torch.bfloat16) for k, v in sd.items()},
                   os.path.join(directory, "model.safetensors"))
        return hf_cfg, sd
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, os.path.join(directory, "model.safetensors"))
    return hf_cfg, sd


def through_the_reference_python(case):
    """(meta dict, state dict): what the reference's Python layer produces for the case -- imports the reference"""
    top = reference_package()
    if top is None:
        raise RuntimeError("needs /root/reference and the built zhilight.C (python -m zhilight_amd.build)")
    sys.path.insert(0, top)
    import torch
    import zhilight                                    # noqa: F401  (the reference's package, against the built C*.so)
    from zhilight import C
    from zhilight.config.adapter import ModelAdapter
    from zhilight.dynamic_batch import DynamicBatchConfig, DynamicBatchGenerator, GeneratorArg
    from zhilight.llama import _get_config
    from zhilight.loader import LLaMALoader
    from zhilight.quant import QuantConfig, quant_config_to_c
    assert os.path.realpath(zhilight.__file__).startswith(os.path.realpath(REFERENCE)), zhilight.__file__
    watched = ("GPTQ_KERNEL_ALGO", "FUSE_GPTQ_MOE", "MOE_DYN_SHARED", "CPM_FUSE_QKV", "CPM_FUSE_FF_IN", "HIGH_PRECISION", "W4_A8_M_THRES",
               "NEED_DEQUANT_WEIGHT", "AWQ_USE_EXLLAMA", "DUAL_STREAM", "W4_INT8_ALGO", "W4_FP8_ALGO")
    saved = {k: os.environ.pop(k, None) for k in watched}
    try:
        with tempfile.TemporaryDirectory() as d:
            hf_cfg, _ = hf_checkpoint(case, d)
            # LLaMA.__init__ (zhilight/llama.py:114-147) up to the first device call
            config = _get_config(ModelAdapter.adapt(dict(hf_cfg)))
            quant = QuantConfig.adapt_hf_config(None, config)
            C.ModelConfig(config)                       # the binding accepts the dict (py_model_config.cpp:77-151)
            cq = quant_config_to_c(quant)
            assert isinstance(cq, C.QuantConfig)
            # LLaMA.load_model_safetensors -> load_state_dict_pt (zhilight/llama.py:186-213)
            force_half = config.get("force_half", False)

            def trans_type(p):
                if p.dtype == torch.bfloat16:
                    return p.half() if force_half else p.view(torch.int16)
                if p.dtype == torch.float8_e4m3fn:
                    return p.view(torch.int8)
                return p
            state = {LLaMALoader._replace_name(name): np.atleast_1d(trans_type(p).cpu().numpy())
                     for name, p in LLaMALoader.load_safetensors(d).items()}
        env = {k: os.environ[k] for k in watched if k in os.environ}
        dc = DynamicBatchConfig(max_batch=4, max_beam_size=1, task_queue_size=8, max_total_token=1024, eos_id=1, bos_id=2, ignore_eos=True).c_config()
        dyn = {f: getattr(dc, f) for f in ("max_batch", "max_beam_size", "task_queue_size", "max_total_token", "seed", "unk_id", "bos_id", "eos_id",
                                           "first_batch", "nccl", "rag_buffer", "ignore_eos", "keep_eos", "reserved_work_mem_mb", "high_precision",
                                           "flash_attention", "enable_prompt_caching")}
        arg = GeneratorArg(beam_size=1, max_length=6)
        arg.session_id = None
        prompt = [int(t) for t in np.random.default_rng(CASES[case]["seed"] + 100).integers(3, CASES[case]["vocab"], 17)]
        task = DynamicBatchGenerator.to_c_task(prompt, arg)
        assert isinstance(task, C.SearchTask) and task.input_tokens_num() == len(prompt)
        task_args = [arg.beam_size, arg.max_length, arg.presence_penalty, arg.repetition_penalty, arg.ngram_penalty,
                     arg.seed is not None and arg.seed != 0, arg.seed or 0, arg.temperature, arg.num_results, arg.top_p, arg.top_k,
                     bool(arg.bee_answer_multi_span), arg.top_logprobs, 0, arg.output_hidden_states]
    finally:
        for k in watched:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
        sys.path.remove(top)
    qv = quant["type"].value if hasattr(quant.get("type"), "value") else int(quant.get("type", 0))
    meta = {"case": case, "config": {k: v for k, v in config.items() if k != "quantization_config"},
            "quantization_config": config.get("quantization_config"),
            "quant_config_to_c": [qv, quant.get("quant_weight_kv", 1), bool(quant.get("act_order", False)), quant.get("group_size", 128),
                                  bool(quant.get("sym", False))],
            "env": env, "dyn_batch_config": dyn, "prompt": prompt, "search_task_args": task_args,
            "state_sha256": {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] for k, v in sorted(state.items())},
            "state_dtypes": {k: str(v.dtype) for k, v in sorted(state.items())},
            "state_shapes": {k: list(v.shape) for k, v in sorted(state.items())}}
    return meta, state


def paths(case):
    return os.path.join(GOLDEN, f"python_layer_{case}.json"), os.path.join(GOLDEN, f"python_layer_{case}.npz")


def main():
    # the extension module's RUNPATH is $ORIGIN, which follows the SYMLINK's directory: give the loader the real directories
    # (LD_LIBRARY_PATH is read at process start, hence the re-exec)
    from zhilight_amd import build
    need = [os.path.dirname(build.binding_target()), os.path.join(ROOT, "zhilight_amd")]
    have = os.environ.get("LD_LIBRARY_PATH", "").split(":")
    if not all(d in have for d in need):
        os.environ["LD_LIBRARY_PATH"] = ":".join(need + [h for h in have if h])
        os.execv(sys.executable, [sys.executable] + sys.argv)
    check = "--check" in sys.argv
    for case in CASES:
        meta, state = through_the_reference_python(case)
        jp, npz = paths(case)
        if check:
            want = json.load(open(jp))
            assert want == json.loads(json.dumps(meta)), case
            with np.load(npz) as z:
                assert sorted(z.files) == sorted(state) and all(np.array_equal(z[k], state[k]) for k in state), case
            print(f"{case}: fixture matches the reference's Python layer ({len(state)} tensors)")
            continue
        os.makedirs(GOLDEN, exist_ok=True)
        with open(jp, "w") as fh:
            json.dump(meta, fh, indent=1, sort_keys=True)
        np.savez_compressed(npz, **state)
        print(f"{case}: wrote {jp} and {npz} ({os.path.getsize(npz) / 1e3:.0f} KB, {len(state)} tensors)")


if __name__ == "__main__":
    main()
