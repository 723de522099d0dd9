"""CPU tests of the host-side mirror (no GPU, no kernels): name mapping, config parsing, bench byte model."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hf_name_mapping_follows_reference_loader():
    from zhilight_amd.llama import hf_name_to_internal as m
    assert m("model.embed_tokens.weight") == "llama.token_embedding.weight"
    assert m("model.norm.weight") == "llama.output_layernorm.weight"
    assert m("lm_head.weight") == "llama.lm_head.weight"
    assert m("model.layers.3.input_layernorm.weight") == "llama.layers.3.ln_attn.weight"
    assert m("model.layers.3.post_attention_layernorm.weight") == "llama.layers.3.ln_ff.weight"
    assert m("model.layers.31.self_attn.q_proj.qweight") == "llama.layers.31.attn.project_q.qweight"
    assert m("model.layers.0.self_attn.o_proj.scales") == "llama.layers.0.attn.attn_out.scales"
    assert m("model.layers.0.mlp.gate_proj.qzeros") == "llama.layers.0.ff.w_in.qzeros"
    assert m("model.layers.0.mlp.up_proj.qweight") == "llama.layers.0.ff.w_gated.qweight"
    assert m("model.layers.0.mlp.down_proj.g_idx") == "llama.layers.0.ff.w_out.g_idx"


def test_model_and_quant_config_from_hf():
    from zhilight_amd._lib import ZLError
    from zhilight_amd.llama import ModelConfig, QuantConfig
    hf = dict(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, num_hidden_layers=32, intermediate_size=14336,
              vocab_size=128256, rms_norm_eps=1e-5, rope_theta=500000.0)
    c = ModelConfig.from_hf(hf)
    assert (c.dim_model, c.num_heads, c.num_kv_heads, c.dim_head, c.dim_ff, c.num_layers) == (4096, 32, 8, 128, 14336, 32)
    assert c == ModelConfig.llama3_8b()
    q = QuantConfig.from_hf(dict(quant_method="gptq", bits=4, group_size=128, sym=True, desc_act=False))
    assert (q.quant_type, q.group_size, q.sym) == (5, 128, True)
    assert QuantConfig.from_hf(dict(quant_method="gptq", bits=4, desc_act=True)).act_order
    a = QuantConfig.from_hf(dict(quant_method="awq", bits=4, group_size=128, zero_point=True, version="gemm"))
    assert (a.quant_type, a.group_size, a.awq, a.sym) == (5, 128, True, False)
    with pytest.raises(ZLError):
        QuantConfig.from_hf(dict(quant_method="fp8"))
    with pytest.raises(ZLError):
        QuantConfig.from_hf(dict(quant_method="gptq", bits=8))


def test_algorithmic_bytes_match_baseline_table():
    """SURVEY 8(d) / BASELINE.md: 3 625 975 808 B of int4 linears per Llama-3-8B decode step."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    per_layer = sum(bench.alg_bytes_w4(n, k, 128, 0) for n, k in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)))
    assert int(per_layer) == 113311744 and int(per_layer) * 32 == 3625975808


def test_baseline_json_is_the_bench_contract():
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "Llama-3-8B GPTQ-Int4 TP=1" in b["metric"] and "batch=1 decode seq=1024" in b["configs"][1]


def test_python_layer_fixture_is_what_the_reference_produces():
    """VERDICT r05 item 3, the CPU half: the reference's OWN Python package (every file of /root/reference/zhilight symlinked into a
    temporary package next to the built `C*.so`; nothing copied or edited) imports against the binding this repository builds, and
    its device-free flow -- config adaptation, quantisation config + environment switches, the safetensors loader's renaming and
    dtype views, DynamicBatchConfig.c_config, to_c_task -- reproduces the committed fixtures tests/golden/python_layer_*.{json,npz}
    bit for bit (tools/gen_python_layer_fixture.py --check, in a child process: the import sets process-wide environment switches).
    The GPU half feeds those fixtures to the same C*.so: tests/test_gpu_zz_binding.py::test_binding_fed_by_the_reference_python_layer."""
    import subprocess
    import sys
    import pytest
    from zhilight_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/zhilight") or not os.path.exists(build.binding_target()):
        pytest.skip("needs the reference tree and the built zhilight.C (this container, after python -m zhilight_amd.build)")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_python_layer_fixture.py"), "--check"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.count("fixture matches the reference's Python layer") == 3, r.stdout[-2000:]   # llama_gptq, ..._desc_act, minicpm_bf16


def test_python_layer_fixture_names_are_the_boundary_loaders():
    """the committed fixtures (the reference loader's HF -> internal renaming) against this repository's own statement of that
    renaming (tests/test_gpu_refcompile.py::_reference_names_state): same names, same bytes -- no reference tree needed"""
    import json
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from test_gpu_refcompile import _reference_names_state
    from tools.gen_python_layer_fixture import CASES, hf_tensors, paths
    for case in CASES:
        jp, npz = paths(case)
        meta = json.load(open(jp))
        _, sd, _ = hf_tensors(case)
        tied = "lm_head.weight" not in sd               # the MiniCPM case: the checkpoint has no lm_head, the loader invents none
        named = dict(sd, **{"lm_head.weight": sd["model.embed_tokens.weight"]}) if tied else sd
        mine = {k.replace("m.", "llama.", 1): v for k, v in _reference_names_state(named).items()}
        if tied:
            del mine["llama.lm_head.weight"]
        with np.load(npz) as z:
            assert sorted(z.files) == sorted(mine) == sorted(meta["state_shapes"])
            for k in mine:
                assert np.array_equal(np.ascontiguousarray(mine[k]).view(np.uint8), z[k].view(np.uint8)), k
        assert meta["quant_config_to_c"][2] == CASES[case]["desc_act"]
        assert (meta["env"].get("GPTQ_KERNEL_ALGO") == "0") == CASES[case]["desc_act"]      # zhilight/quant.py:73-76
