// host_attention_ext.cpp -- the names attention.cpp references next to the boundary's operators: nn::FlashDecoding::mha_fwd (the
// prompt attention of attn_encode_group, attention.cpp:553-562) over zl_prefill_attn; FlashDecoding's varlen / compact entry
// points, the unfused softmax route and the prompt encode without flash attention stay off the path: definitions that throw.
#include <cmath>
#include <vector>

#include "host_common.h"
#include "nn/attention/attention_kernel.h"
#include "nn/attention/flash_decoding.h"

namespace nn {

FlashDecoding::FlashDecoding(const Context&) {}
FlashDecoding::~FlashDecoding() = default;
core::Tensor FlashDecoding::forward(const Context&, Tensor&, const Tensor&, const Tensor&, Tensor*, const Tensor*, const Tensor*, int, int, bool, bool, int, int,
                                    float) {
    ZL_OFF_PATH("nn::FlashDecoding::forward (USE_FA_DECODING; the ragged decode takes multi_query_attention_rag_buffer)");
}
core::Tensor FlashDecoding::compact_kv_fwd(const Context&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                           const Tensor*, Tensor, float) {
    ZL_OFF_PATH("nn::FlashDecoding::compact_kv_fwd (dynamic batching off)");
}
// the prompt ("encode") attention of one task: q (1, n, H, D) against the first len_kv rows of its buffers (1, len_kv, Hkv, D), which
// already hold the chunk's own rows; causal with the chunk at the END of the keys (flash-attn's bottom-right alignment) =
// zl_prefill_attn with pos0 = len_kv - n (attention.cpp:553-562 is the only call site)
core::Tensor FlashDecoding::mha_fwd(const Context& ctx, Tensor& q, const Tensor& k, const Tensor& v, Tensor* out_, Tensor* alibi_slopes, const float p_dropout,
                                    const float softmax_scale, bool is_causal, int window_size_left, int window_size_right, const float softcap,
                                    const bool return_softmax) {
    BM_ASSERT(q.ndim() == 4 && k.ndim() == 4 && q.size(0) == 1 && k.size(0) == 1, "mha_fwd: (1, len, heads, dim) operands");
    if (!is_causal || alibi_slopes || p_dropout != 0.f || window_size_left >= 0 || window_size_right >= 0 || softcap != 0.f || return_softmax)
        ZL_OFF_PATH("nn::FlashDecoding::mha_fwd with anything but plain causal attention");
    const int64_t n = q.size(1), h = q.size(2), d = q.size(3), len_kv = k.size(1), hkv = k.size(2);
    Tensor out = out_ ? *out_ : ctx.tensor(q.shape(), q.dtype());
    if (d != 128) {
        // head sizes the matrix-core prompt kernel does not take (MiniCPM: 64): the mask form of zl_decode_attn -- one task, n query
        // rows, key t visible to row i iff t <= len_kv - n + i -- as zhilight_amd/llama.py::_prefill_mask does for the Python driver
        std::vector<int8_t> m((size_t)n * len_kv);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t t = 0; t < len_kv; ++t) m[(size_t)i * len_kv + t] = t <= len_kv - n + i;
        Tensor mask = ctx.tensor_of(m, {(size_t)n * (size_t)len_kv}, DataType::kInt8);
        Tensor lens = ctx.tensor_of(std::vector<int32_t>{(int32_t)len_kv}, {1}, DataType::kInt32);
        auto ptr_table = [&](const Tensor& t) {      // a one-entry device pointer table (two int32 words: the shim has no 64-bit dtype)
            const uint64_t a = (uint64_t)(uintptr_t)t.data();
            return ctx.tensor_of(std::vector<int32_t>{(int32_t)(uint32_t)a, (int32_t)(uint32_t)(a >> 32)}, {2}, DataType::kInt32);
        };
        Tensor ka = ptr_table(k), va = ptr_table(v);
        const int64_t wbytes = zl_decode_attn_workspace_bytes(1, n, h, d, len_kv);
        BM_ASSERT(wbytes > 0, "mha_fwd: attention workspace");
        Tensor ws = ctx.tensor({(size_t)(wbytes + 3) / 4}, DataType::kFloat);
        ZL_CK(zl_decode_attn(q.data<uint16_t>(), lens.data<int32_t>(), reinterpret_cast<const uint16_t* const*>(ka.data()), reinterpret_cast<const uint16_t* const*>(va.data()),
                             mask.data<int8_t>(), nullptr, out.data<uint16_t>(), ws.data(), 1, n, h, hkv, d,
                             softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)d), len_kv, 1,
                             q.dtype() == DataType::kHalf ? ZL_F16 : ZL_BF16, (zl_stream_t)ctx.current_cuda_stream()),
              "mha_fwd (masked attention, head size != 128)");
        return out;
    }
    ZL_CK(zl_prefill_attn(q.data<uint16_t>(), k.data<uint16_t>(), v.data<uint16_t>(), out.data<uint16_t>(), n, len_kv - n, h, hkv, d,
                          softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)d), len_kv, 1, q.dtype() == DataType::kHalf ? ZL_F16 : ZL_BF16,
                          (zl_stream_t)ctx.current_cuda_stream()),
          "prefill_attn");
    return out;
}
void attn_softmax(const core::Context&, float, const core::Tensor&, const core::Tensor&, const core::Tensor&) {
    ZL_OFF_PATH("nn::attn_softmax (the unfused gemm + softmax + gemm route)");
}
void multi_query_self_attention(const core::Context&, const core::Tensor&, const core::Tensor&, const core::Tensor&, const core::Tensor&, float, core::Tensor&, int) {
    ZL_OFF_PATH("nn::multi_query_self_attention (prompt encode without flash attention)");
}

}  // namespace nn
