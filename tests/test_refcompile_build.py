"""VERDICT r02 item 6, the CPU half: the reference's src/nn/linear/linear.cpp compiles UNMODIFIED against the boundary's
headers (hostcpp/refshim + bm_hip.h / bm_layer.h / bm_functions.h) and every name it references is defined by the
boundary (zhilight_amd.build.build_refcompile fails on a leftover undefined symbol).  Needs the reference tree, i.e. this
container; elsewhere the prebuilt module (if it travelled) is only imported."""
import os
import sys

import pytest


def test_reference_linear_tu_builds_and_links_against_the_boundary():
    from zhilight_amd import _lib, build
    build.build()                   # no-op when up to date (tests/test_abi.py does the same)
    _lib.lib()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_TUS)
    path = build.build_refcompile() if have_reference else build.refcompile_target()
    if not (path and os.path.exists(path)):
        pytest.skip("no reference tree and no prebuilt module")
    sys.path.insert(0, os.path.dirname(path))
    try:
        import zl_reflinear
    finally:
        sys.path.pop(0)
    assert hasattr(zl_reflinear, "RefLinear") and zl_reflinear.weight_cache_size() == 0


def test_reference_mla_tu_compiles_and_calls_the_boundary_by_its_own_signatures():
    """The reference's MLA attention layer (src/nn/attention/multi_head_latent_attention.cpp, 1 500 lines) compiles unmodified
    against the shim, and every name it references in the namespaces the boundary stands in for -- ds:: (the FlashMLA binding),
    bmengine:: (tensor / context / functions), nn::fp8 / nn::gptq / int8_op -- is DEFINED by the boundary under the same mangled
    name, i.e. with the reference's exact signature (build_refcheck raises otherwise).  What stays outside is the device code of
    layers that are not on this path (their .cu files): a short, known list."""
    import json
    from zhilight_amd import build
    build.build()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_CHECK_TUS)
    report = build.build_refcheck() if have_reference else build.refcheck_report()
    if not (report and os.path.exists(report)):
        pytest.skip("no reference tree and no prebuilt report")
    v = json.load(open(report))["src/nn/attention/multi_head_latent_attention.cpp"]
    resolved, outside = v["resolved"], v["outside"]
    for name in ("ds::mha_fwd_kvcache_mla(", "ds::get_mla_metadata(", "bmengine::functions::copy_last_dim(", "bmengine::functions::concat_broadcast_b(",
                 "bmengine::core::Context::get_cache_allocator(", "nn::Linear::forward(",
                 "nn::multi_query_attention_rag_buffer(", "nn::copy_to_rag_buffer2("):
        assert any(n.startswith(name) for n in resolved), name
    assert not [n for n in outside if n.startswith(build.REF_CHECK_NAMESPACES)]
    owners = {n.split("(")[0].rsplit("::", 1)[0] if n.split("(")[0].count("::") > 1 else n.split("(")[0] for n in outside}
    assert owners <= {"nn::FlashDecoding", "nn::RotaryEmbedding", "nn::LayerNorm", "model::ModelContext", "kvcache::TransformerBuffer",
                      "kvcache::copy_to_buffer", "nn::attn_softmax", "nn::copy_to_rag_buffer"}, owners


def test_reference_attention_tu_binds_the_decode_hot_path_names():
    """src/nn/attention/attention.cpp -- NormalImpl::dynamic_batch_forward, the decode hot path itself -- compiled
    unmodified: the operators the boundary stands in for (fused-attention kernel, its workspace, rotary variants, KV scatter, the
    INT8-KV quantiser, Linear, LayerNorm, the bmengine functions it uses) are defined under the reference's signatures; outside stay
    the classes whose device code is the reference's own (.cu / borrowed libraries)."""
    import json
    from zhilight_amd import build
    build.build()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_CHECK_TUS)
    report = build.build_refcheck() if have_reference else build.refcheck_report()
    if not (report and os.path.exists(report)):
        pytest.skip("no reference tree and no prebuilt report")
    v = json.load(open(report))["src/nn/attention/attention.cpp"]
    for name in ("nn::multi_query_attention_rag_buffer(", "nn::attention_qkv_rag_buffer(", "nn::get_mqa_workspace(", "nn::rope_qk_cache(", "nn::rotary_embedding_qk(",
                 "nn::copy_to_rag_buffer2(", "int8_op::quant_calc_scale(", "nn::Linear::forward(", "nn::Linear::fuse(", "nn::LayerNorm::forward(",
                 "bmengine::functions::Gemm::forward(", "bmengine::functions::transpose_2_1(", "bmengine::core::Context::get_allocator("):
        assert any(n.startswith(name) for n in v["resolved"]), name
    assert not v["pending"] and not [n for n in v["outside"] if n.startswith(build.REF_CHECK_NAMESPACES)]
    # round 4: the unit is linked into the executed test module (zl_reflinear: hostcpp/ref_attention_glue.cpp provides the KV buffer
    # class, RotaryEmbedding, FlashDecoding::mha_fwd over zl_prefill_attn, ModelContext's constructor) -- nothing is left outside,
    # and tests/test_gpu_refcompile.py RUNS its decode and encode paths
    assert not v["outside"], v["outside"]
    for name in ("kvcache::TransformerBuffer::copy(", "nn::RotaryEmbedding::forward(", "nn::FlashDecoding::mha_fwd(", "nn::Attention::impl::create_mla_impl("):
        assert any(n.startswith(name) for n in v["resolved"]), name


def test_reference_block_tu_binds_the_layer_orchestration_names():
    """src/nn/block/block.cpp (EncoderLayer::forward, single_stream_encode, dual_stream_encode: SURVEY 8a row a19): the residual adds,
    the fused add + norm, the stream / allocator switching of the dual-stream prompt path and bmengine's elementwise glue bind to the
    boundary; Attention and FeedForward are the reference's own classes from the units checked above; the tensor-parallel reduces go
    through ModelContext (the reference's model_context.cpp, outside this check)."""
    import json
    from zhilight_amd import build
    build.build()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_CHECK_TUS)
    report = build.build_refcheck() if have_reference else build.refcheck_report()
    if not (report and os.path.exists(report)):
        pytest.skip("no reference tree and no prebuilt report")
    v = json.load(open(report))["src/nn/block/block.cpp"]
    for name in ("nn::element_add_scale_out(", "nn::LayerNorm::fuse_add(", "nn::LayerNorm::forward(", "bmengine::core::Context::use_cache_alloc(",
                 "bmengine::core::Context::reserve_cache_alloc(", "bmengine::core::Context::set_current_stream(",
                 "bmengine::functions::BinaryElementwiseOp::forward(", "bmengine::functions::reduce_abs_max("):
        assert any(n.startswith(name) for n in v["resolved"]), name
    # round 4: block.cpp, feedforward.cpp and attention.cpp are all in the executed module (tests/test_gpu_refcompile.py runs
    # EncoderLayer::forward): the reference's own classes resolve against each other's units
    for name in ("nn::FeedForward::forward(", "nn::Attention::forward("):
        assert any(n.startswith(name) for n in v["resolved"] + v["reference"]), name
    assert {n.split("(")[0] for n in v["pending"]} <= {"bmengine::functions::pow", "bmengine::functions::clamp"}
    assert not v["outside"], v["outside"]     # (round 3: four ModelContext members and two of LayerNorm; hostcpp/ref_block_glue.cpp provides them)


def test_reference_feedforward_tu_binds_the_router_dispatch_and_fp8_names():
    """The same check on src/nn/feedforward/feedforward.cpp (1 290 lines: dense and MoE feed-forward incl. the dispatch route):
    the router, the dispatch / combine helpers of ff_kernel.h, nn::fp8::per_token_cast_to_fp8, the fused GPTQ MoE GEMVs, the grouped
    FP8 GEMM of Linear and c10d::NCCLBroadcast are all defined by the boundary under the reference's signatures.  (Rounds 1-3: four helpers of
    bmengine's functions library -- arange, sort_pair_1d, divide, scatter_update_dim0 -- were pending; round 4 provides them as kernels and
    the unit RUNS, tests/test_gpu_refcompile.py.  The pending list may only shrink.)"""
    import json
    from zhilight_amd import build
    build.build()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_CHECK_TUS)
    report = build.build_refcheck() if have_reference else build.refcheck_report()
    if not (report and os.path.exists(report)):
        pytest.skip("no reference tree and no prebuilt report")
    v = json.load(open(report))["src/nn/feedforward/feedforward.cpp"]
    for name in ("nn::top_k_softmax(", "nn::group_topk_softmax(", "nn::sum_experts(", "nn::route_shared_lb(", "nn::plus_for_sort(",
                 "nn::calc_reverse_idx(", "nn::fill_m_indices_padded_indices(", "nn::fp8::per_token_cast_to_fp8(", "nn::gptq::gemm_moe_up(",
                 "nn::gptq::gemm_moe_down(", "nn::gptq::gemm_fuse_gate_in(", "nn::gate_mul_inplace(", "nn::gate_fuse(", "nn::Linear::grouped_gemm_fp8_block(",
                 "bmengine::c10d::NCCLBroadcast(", "bmengine::functions::index_select("):
        assert any(n.startswith(name) for n in v["resolved"]), name
    assert {n.split("(")[0] for n in v["pending"]} <= {"bmengine::functions::arange", "bmengine::functions::sort_pair_1d",
                                                       "bmengine::functions::divide", "bmengine::functions::scatter_update_dim0"}
    assert not v["outside"]                                                  # every other name of the unit is defined by the boundary


def test_refshim_holds_no_reference_text():
    """the shim directory is this repository's own code: forwarding headers + aliases, no copied reference header"""
    shim = os.path.join(os.path.dirname(__file__), "..", "zhilight_amd", "hostcpp", "refshim")
    for root, _, files in os.walk(shim):
        for f in files:
            text = open(os.path.join(root, f)).read()
            assert "BMENGINE_EXPORT" not in text, f
            assert len(text.splitlines()) < 200, f


def test_reference_binding_links_and_imports():
    """Round 5: the reference's own `zhilight.C` -- src/py_export/{bind,py_batch_generator,py_llama,py_model_base,py_model_config,py_utils}.cpp and the
    dynamic-batch scheduler src/generator/batch_generator.cpp, every unit compiled UNMODIFIED -- links on libzhilight_amd_host.so with no name left
    undefined (zhilight_amd.build.build_binding raises otherwise), imports without a GPU, and carries the surface zhilight/llama.py and
    zhilight/dynamic_batch.py use.  tests/test_gpu_zz_binding.py runs it on the MI355X."""
    from zhilight_amd import _lib, build
    build.build()
    _lib.lib()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_BINDING_TUS)
    path = build.build_binding() if have_reference else build.binding_target()
    if not (path and os.path.exists(path)):
        pytest.skip("no reference tree and no prebuilt module")
    assert os.path.exists(build.host_target())
    sys.path.insert(0, os.path.dirname(path))
    try:
        import C
    finally:
        sys.path.pop(0)
    for cls, methods in (("Engine", ()), ("ModelConfig", ()), ("QuantConfig", ()), ("DistConfig", ()), ("CPMBase", ()),
                         ("LLaMA", ("load_state_dict", "load_with_smooth_quant", "calc_act_scales", "get_input_embeddings")),
                         ("DynBatchConfig", ("max_batch", "max_beam_size", "rag_buffer", "flash_attention", "eos_id", "bos_id")),
                         ("SearchTask", ("get_result", "has_result", "cancel", "set_logit_bias", "input_tokens_num")),
                         ("BatchGenerator", ("run", "stop", "submit", "batch_search", "queue_size", "active_size"))):
        assert hasattr(C, cls), cls
        for m in methods:
            assert hasattr(getattr(C, cls), m), (cls, m)
    # host-side objects that need no device: the config classes and a task
    cfg = C.ModelConfig({"model_type": "llama", "num_layers": 2, "dim_model": 1024, "num_heads": 8, "dim_head": 128, "dim_ff": 2048, "vocab_size": 512,
                         "eps": 1e-5, "num_kv_heads": 2, "dtype": "half"})
    assert cfg is not None and C.QuantConfig(5, True, False, 128, False) is not None
    task = C.SearchTask([5, 6, 7], 1, 4, 0.0, 1.0, 1.0, False, 0, 1.0, 1, 1.0, 0, False, 0, 0, 0)
    assert task.input_tokens_num() == 3 and not task.has_result()
    # and the link report of the binding's units: every one compiles, nothing is left outside the boundary
    import json
    report = build.build_refcheck() if have_reference else build.refcheck_report()
    if report and os.path.exists(report):
        r = json.load(open(report))
        for rel in build.REF_REPORT_TUS:
            if rel in r:
                assert r[rel]["compiles"] and not r[rel]["outside"], (rel, r[rel].get("outside"), r[rel].get("first_errors"))


def test_host_library_carries_the_engine_the_host_classes_and_the_reference_units():
    """libzhilight_amd_host.so (round 5): one library with core::Engine, the classes the reference keeps in .cu files, and the reference's own
    host units -- among them model_context.cpp, so ModelContext::create / reduce_sum / reduce_sum2 are the reference's code -- every name resolved
    against libzhilight_amd.so / libzhilight_amd_comm.so (build_host raises otherwise)."""
    import subprocess
    from zhilight_amd import build
    build.build()
    have_reference = all(os.path.exists(os.path.join(build.REFERENCE, t)) for t in build.REF_TUS)
    path = build.build_host() if have_reference else build.host_target()
    if not (path and os.path.exists(path)):
        pytest.skip("no reference tree and no prebuilt library")
    syms = subprocess.check_output(["nm", "-D", "--defined-only", "-C", path], text=True)
    for name in ("bmengine::core::Engine::create_context_rank(int) const", "bmengine::core::Engine::device_foreach(",
                 "model::ModelContext::create(", "model::ModelContext::reduce_sum(", "model::ModelContext::reduce_sum2(", "model::ModelContext::reduce_tp_int8(",
                 "model::LLaMA::encode(", "nn::EncoderLayer::forward(", "nn::Attention::dyn_rag_forward(", "nn::FeedForward::forward(", "nn::Linear::forward(",
                 "kvcache::TransformerBuffer::resize(", "kvcache::TransformerBuffer::dump_slice(", "nn::RotaryEmbedding::rotate(", "nn::RopePreparer::forward(",
                 "nn::RawEmbedding::projection(", "nn::LayerNorm::forward_2(", "nn::FlashDecoding::mha_fwd(", "bmengine::functions::TopK::forward(",
                 "beam_utility::log_softmax_bias(", "beam_utility::random_sampler_gpu(", "bmengine::c10d::NCCLAllReduce(", "int8_op::quant_group_32(",
                 "deep_gemm_fp8_block_h20_group", "curandGenerateUniform"):
        assert name in syms, name
    needed = subprocess.check_output(["readelf", "-d", path], text=True)
    assert "libzhilight_amd.so" in needed and "libzhilight_amd_comm.so" in needed
    assert "libcudart" not in needed and "libnccl" not in needed and "libcublas" not in needed
