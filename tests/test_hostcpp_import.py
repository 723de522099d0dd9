"""CPU checks of the C++ host layer (zhilight_amd/hostcpp): it builds, the pybind test module imports without a GPU and
exposes the operator entry points the GPU tests drive; every nn:: / int8_op:: name declared in nn_amd.h is defined in the
shared object (an undefined one would fail at import: the module is linked with the full wrapper set)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hostcpp_builds_and_imports():
    from zhilight_amd import build as zbuild
    zbuild.build()
    assert os.path.exists(zbuild.hostcpp_target())
    from zhilight_amd import _lib
    _lib.lib()
    from zhilight_amd import zl_internals as zi
    for name in ("gptq_gemm_k_major", "gptq_dequant_k_major", "gemm_fuse_gate_in", "gptq_load_transforms",
                 "multi_query_attention_rag_buffer", "rope_qk_cache", "copy_to_rag_buffer2", "layernorm", "layernorm_fuse_add",
                 "element_add_scale", "gate_mul", "quant_calc_scale", "layernorm_quant", "int8_linear", "quant_back_act_mul"):
        assert hasattr(zi.Context, name), name


def test_wrapper_names_declared_are_defined():
    from zhilight_amd import build as zbuild
    so = zbuild.build_hostcpp()
    syms = subprocess.run(["nm", "-DC", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    hdr = open(os.path.join(ROOT, "zhilight_amd", "hostcpp", "nn_amd.h")).read()
    # free functions of the three namespaces, by name
    wanted = {"nn::gptq::gptq_gemm_k_major", "nn::gptq::gemm_fuse_gate_in", "nn::gptq::dequant_k_major", "nn::gptq::gptq_shuffle",
              "nn::gptq::increase_zero", "nn::gptq::q4_to_q8", "nn::gptq::un_shuffle", "nn::gptq::shuffle_awq", "nn::gptq::amd_pack_k_major",
              "nn::get_mqa_workspace", "nn::multi_query_attention_rag_buffer", "nn::rotary_embedding_qk", "nn::rope_qk_cache",
              "nn::copy_to_rag_buffer2", "nn::element_add_scale", "nn::element_add_scale_out", "nn::gate_mul_inplace",
              "nn::LayerNorm::forward", "nn::LayerNorm::fuse_add", "nn::LayerNorm::inplace",
              "int8_op::quant_calc_scale", "int8_op::quant_scale_back", "int8_op::quant_scale_back3", "int8_op::layernorm_quant",
              "int8_op::quant_back_element_add_scale", "int8_op::quant_back_transpose", "int8_op::quant_back_act_mul",
              "int8_op::quant_back_copy_to_buffer", "int8_op::int8_gemm_nt",
              "bmengine::core::Context::tensor", "bmengine::core::Tensor::slice_dim0", "bmengine::core::Tensor::from_external"}
    for w in wanted:
        assert re.search(r"\b" + re.escape(w) + r"\(", syms), w
        assert w.split("::")[-1] in hdr or w.startswith("bmengine"), w
