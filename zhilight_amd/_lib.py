"""ctypes loader for libzhilight_amd.so (the gfx950 C-ABI library, include/zhilight_amd.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved this raises,
so a GPU box can never silently run anything but the HIP path.
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime of the process BEFORE our library
# (linked against the same soname) is dlopen'ed, otherwise two runtimes coexist and torch's streams
# and allocations are foreign to our launches ("no ROCm-capable device").  torch is the plumbing for
# device memory and streams on this path, so importing it here is not optional.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZHILIGHT_AMD_SO: load an experimental build of the same library instead (tools/ubench/variant.sh)
SO_PATH = os.environ.get("ZHILIGHT_AMD_SO") or os.path.join(_HERE, "libzhilight_amd.so")

# every entry point declared in include/zhilight_amd.h (kept in sync by tests/test_abi.py)
SYMBOLS = [
    "zl_mla_decode_workspace_bytes", "zl_mla_decode_attn", "zl_mla_decode_attn_ex", "zl_mla_decode_attn_paged",
    "zl_moe_sum_experts", "zl_moe_sum_experts_arr", "zl_moe_route_shared_lb", "zl_moe_plus_for_sort", "zl_moe_calc_reverse_idx",
    "zl_moe_fill_m_indices",
    "zl_embedding_rope",
    "zl_fp8_per_token_cast", "zl_fp8_block_dequant", "zl_fp8_block_gemm_group", "zl_fp8_block_packed_bytes", "zl_fp8_block_pack", "zl_fp8_block_gemm_group_packed", "zl_moe_top_k_softmax", "zl_moe_group_topk",
    "zl_cast", "zl_copy_2d", "zl_index_select", "zl_argmax_advance", "zl_arange_i32", "zl_divide_i32", "zl_scatter_update_dim0", "zl_sort_pairs_i32", "zl_log_softmax_bias", "zl_softmax_rows", "zl_topk_rows", "zl_gather_logits", "zl_scatter_logits", "zl_repetition_penalty", "zl_reduce_abs_max", "zl_binary_op", "zl_scale", "zl_act_inplace",
    "zl_count_nonfinite", "zl_perm_narrow_u16", "zl_perm_reverse_u16", "zl_permute_input_u16", "zl_gptq_permute_rows",
    "zl_version", "zl_status_string", "zl_device_cu_count",
    "zl_gptq_shuffle", "zl_gptq_increase_zero", "zl_gptq_q4_to_q8", "zl_transpose_2d", "zl_gptq_reconstruct",
    "zl_awq_un_shuffle", "zl_awq_shuffle",
    "zl_w4_layout", "zl_w4_pack", "zl_w4_dequant", "zl_w4a16_gemm",
    "zl_w4m_layout", "zl_w4m_pack", "zl_w4m_unpack", "zl_w4a16_gemm_mfma", "zl_w4a16_gemm_tiled", "zl_w4a16_scratch_bytes", "zl_w4a16_gemm_mfma_ex", "zl_w4a16_gemm_tiled_ex",
    "zl_gemm_nt_small_m", "zl_gemm_nt", "zl_gemm_nt_f32", "zl_argmax_workspace_bytes", "zl_gemm_nt_small_m_argmax", "zl_greedy_advance",
    "zl_rmsnorm",
    "zl_w4a16_moe_up", "zl_w4a16_moe_down", "zl_rope_cos_sin", "zl_rope_cos_sin_llama3", "zl_rope_cos_sin_dynamic", "zl_rope_cos_sin_yarn", "zl_head_norm", "zl_rotary_embedding_qk", "zl_rope_qk_cache", "zl_rope_rotate", "zl_mask_valid_lens",
    "zl_copy_to_rag_buffer2", "zl_copy_to_rag_buffer_bytes", "zl_rope_scatter_decode", "zl_w4a16_qkv_rope_scatter", "zl_w4a16_qkv_rope_scatter_ex", "zl_decode_attn_splits_h", "zl_decode_attn_splits_h_mask", "zl_decode_attn_combine_h", "zl_w4a16_gemm_attn_merge_h", "zl_w4a16_gemm_attn_merge_h_ex",
    "zl_quant_group_32", "zl_dequant_sum_quant_g32", "zl_dequant_group_32",
    "zl_fp8_calc_scale", "zl_fp8_cvt_half", "zl_fp8_gemm_nt",
    "zl_decode_attn_workspace_bytes", "zl_decode_attn", "zl_decode_attn_ex", "zl_decode_attn_fused",
    "zl_decode_attn_split_len", "zl_decode_attn_splits", "zl_w4a16_gemm_attn_merge",
    "zl_decode_attn_la_split_len", "zl_decode_attn_la_workspace_bytes", "zl_decode_attn_la",
    "zl_quant_calc_scale_zp", "zl_dequant_group", "zl_quant_copy_to_rag_buffer", "zl_rope_quant_scatter_decode", "zl_decode_attn_quant", "zl_decode_attn_quant_ex",
    "zl_prefill_attn", "zl_prefill_attn_ex",
    "zl_element_add_scale", "zl_gate_mul", "zl_gate_fuse", "zl_dense_m_bytes", "zl_dense_pack_m", "zl_gemm_nt_packed", "zl_row_ss", "zl_w4a16_emits_row_ss", "zl_w4a16_takes_row_ss", "zl_permute_input", "zl_embedding",
    "zl_w8m_bytes", "zl_w8m_pack", "zl_w8a8_gemm_phase", "zl_w8a8_gemm_phase_ex", "zl_w8a8_qkv_rope_scatter",
    "zl_quant_calc_scale", "zl_rmsnorm_quant", "zl_int8_gemm_nt", "zl_quant_scale_back",
    "zl_quant_back_act_mul", "zl_quant_scale_back3", "zl_quant_back_element_add_scale", "zl_quant_back_transpose",
    "zl_quant_back_copy_to_buffer",
    "zl_w4a8_weight_to_int8", "zl_quant_scale_back_f32",
    "zl_awq_dequantize", "zl_awq_gemm_workspace_bytes", "zl_awq_gemm",
]


# entry points of a ZL_BUILD_EXPERIMENTAL=1 build only (include/zhilight_amd.h, #ifdef ZL_EXPERIMENTAL): the digit-plane route for
# 5..32 rows and the loader / consumer engine's fused launch -- built, exact, measured not faster, off the product path
EXPERIMENTAL_SYMBOLS = ["zl_w4a16_attn_out_gate_up", "zl_engine_epoch_advance",
                        "zl_w4a16_planes_bytes", "zl_w4a16_planes", "zl_w4a16_gemm_planes", "zl_w4a16_qkv_rope_scatter_planes"]
experimental = False      # set by lib(): the loaded library exports all of them


class W4Opts(C.Structure):
    """zl_w4_opts_t: caller-provided scratch + explicit tuning overrides of the W4A16 matrix-core launchers"""
    _fields_ = [("scratch", C.c_void_p), ("scratch_bytes", C.c_int64)] + [(n, C.c_int) for n in (
        "phase_rounds", "phase_ksplit", "phase_ksplit_min_m", "phase_min_m", "phase_max_m", "phase_small_off",
        "tiled_min_m", "tiled_bm", "tiled_splitk", "mfma_ks", "mfma_rounds", "small_algo", "tiled_wide",
        "slab", "slab_min_m", "slab_nw", "slab_gpw", "slab_r", "defer_norm")] + [("row_ss", C.c_void_p), ("row_ss_out", C.c_void_p)]


class W4Layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("n", "k", "group_size", "np", "kp", "q", "c", "qw_bytes", "scales_bytes", "zeros_bytes")]


_lib = None


def lib():
    """Load (once) and return the C-ABI library; raises if it is absent -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} not found: build it with `python -m zhilight_amd.build` "
                "(hipcc --offload-arch=gfx950). zhilight_amd has no CPU/PyTorch fallback.")
        l = C.CDLL(SO_PATH)
        for name in SYMBOLS:
            getattr(l, name)  # AttributeError if the library does not export it
        l.zl_status_string.restype = C.c_char_p
        l.zl_decode_attn_workspace_bytes.restype = C.c_int64
        l.zl_decode_attn_split_len.restype = C.c_int64
        l.zl_decode_attn_la_split_len.restype = C.c_int64
        l.zl_decode_attn_la_workspace_bytes.restype = C.c_int64
        l.zl_argmax_workspace_bytes.restype = C.c_int64
        l.zl_w8m_bytes.restype = C.c_int64
        l.zl_dense_m_bytes.restype = C.c_int64
        l.zl_fp8_block_packed_bytes.restype = C.c_int64
        l.zl_mla_decode_workspace_bytes.restype = C.c_int64
        l.zl_w4a16_scratch_bytes.restype = C.c_int64
        global experimental
        experimental = all(hasattr(l, name) for name in EXPERIMENTAL_SYMBOLS)
        if experimental:
            l.zl_w4a16_planes_bytes.restype = C.c_int64
        l.zl_awq_gemm_workspace_bytes.restype = C.c_int64
        _lib = l
    return _lib


COMM_SO_PATH = os.path.join(_HERE, "libzhilight_amd_comm.so")
COMM_SYMBOLS = [
    "zl_comm_unique_id", "zl_comm_create", "zl_comm_destroy", "zl_comm_rank", "zl_comm_size", "zl_comm_all_reduce_sum",
    "zl_comm_all_gather", "zl_comm_reduce_scatter_sum", "zl_comm_broadcast", "zl_comm_send", "zl_comm_recv",
    "zl_comm_group_start", "zl_comm_group_end",
    "zl_ar_buffer_bytes", "zl_ar_state_bytes", "zl_ar_alloc", "zl_ar_free", "zl_ar_export", "zl_ar_open", "zl_ar_close",
    "zl_ar_init", "zl_ar_all_reduce", "zl_ar_all_reduce_int8", "zl_ar_status",
]
_comm = None


def comm_lib():
    """Load (once) libzhilight_amd_comm.so (include/zhilight_amd_comm.h): RCCL communicator + one-shot all-reduce."""
    global _comm
    if _comm is None:
        if not os.path.exists(COMM_SO_PATH):
            raise ImportError(f"{COMM_SO_PATH} not found: build it with `python -m zhilight_amd.build`")
        l = C.CDLL(COMM_SO_PATH)
        for name in COMM_SYMBOLS:
            getattr(l, name)
        l.zl_ar_buffer_bytes.restype = C.c_int64
        l.zl_ar_state_bytes.restype = C.c_int64
        _comm = l
    return _comm


class ZLError(RuntimeError):
    """Raised for a non-zero status of a C-ABI call (the reference raises BMEngineException ->
    Python RuntimeError for the same conditions)."""


def check(status, what):
    if status != 0:
        msg = lib().zl_status_string(C.c_int(status)).decode()
        raise ZLError(f"{what}: status {status}: {msg}")
