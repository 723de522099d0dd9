// host_kvcache.cpp -- kvcache::KVCache / TransformerBuffer (src/kvcache/transformer_buffer.h:11-63; the reference implements them
// in transformer_buffer.cu:267-392): the per-task, per-layer K / V buffers model::RagBufferContext (header-only,
// rag_buffer_context.h) hands out addresses of.  On core::Context::tensor: resize keeps the rows written so far; the prompt's
// rows enter through zl_copy_to_rag_buffer2, the INT8 cache through zl_quant_calc_scale_zp / zl_dequant_group.  Also the paged
// cache's names (src/kvcache/paged_kvcache.h), which buffer_context.cpp references and the ragged-buffer path never constructs:
// definitions that throw.
#include "host_common.h"
#include "kvcache/paged_kvcache.h"
#include "kvcache/transformer_buffer.h"

namespace kvcache {

KVCache::KVCache(int batch_size, int num_layers, int num_heads, int dim_head, core::DataType dtype, bool parallel, bool BSHD)
    : batch_size(batch_size), num_layers(num_layers), num_heads(num_heads), dim_head(dim_head), dtype(dtype), parallel(parallel), BSHD(BSHD) {}

TransformerBuffer::TransformerBuffer(int batch_size, int num_layers, int num_heads, int dim_head, core::DataType dtype, bool parallel, bool BSHD)
    : KVCache(batch_size, num_layers, num_heads, dim_head, dtype, parallel, BSHD) {
    buffer.resize(num_layers);
    scales_.resize(num_layers);
}
TransformerBuffer::TransformerBuffer(const KVCacheConfig& c)
    : TransformerBuffer(-1, c.num_layers, c.num_heads, c.dim_head, c.dtype, false, c.BSHD) {
    scale_dtype_ = c.scale_dtype;
    layer_devices = c.layer_devices;
}
TransformerBuffer::~TransformerBuffer() = default;

void TransformerBuffer::check_layer(int i) const {
    BM_ASSERT(i >= 0 && (size_t)i < num_layers, "TransformerBuffer: layer out of range");
}
const core::Tensor& TransformerBuffer::operator[](int i) const { check_layer(i); return buffer[i]; }
core::Tensor& TransformerBuffer::operator[](int i) { check_layer(i); return buffer[i]; }
const core::Tensor& TransformerBuffer::get_scale(int i) const { check_layer(i); return scales_[i]; }

// grow every layer's buffer to new_length rows, keeping what has been written (per-task buffers: batch_size == -1;
// (len, heads, dim) under BSHD, (heads, len, dim) otherwise)
void TransformerBuffer::resize(const core::Context& ctx, size_t new_length) {
    BM_ASSERT(is_dyn_batch(), "TransformerBuffer: only the per-task (ragged) form is provided here");
    hipStream_t st = ctx.current_cuda_stream();
    auto grow = [&](core::Tensor& old, size_t row_elems, core::DataType dt) {
        const size_t esz = core::get_elem_size(dt);
        const size_t old_len = old.numel() ? old.size(BSHD ? 0 : 1) : 0;
        if (old_len >= new_length) return;
        core::Tensor nw = BSHD ? ctx.tensor({new_length, num_heads, row_elems}, dt) : ctx.tensor({num_heads, new_length, row_elems}, dt);
        BM_CUDART_ASSERT(hipMemsetAsync(nw.data(), 0, nw.nbytes(), st));
        if (old_len) {
            if (BSHD) {
                BM_CUDART_ASSERT(hipMemcpyAsync(nw.data(), old.data(), old.nbytes(), hipMemcpyDeviceToDevice, st));
            } else {
                BM_CUDART_ASSERT(hipMemcpy2DAsync(nw.data(), new_length * row_elems * esz, old.data(), old_len * row_elems * esz,
                                                  old_len * row_elems * esz, num_heads, hipMemcpyDeviceToDevice, st));
            }
        }
        BM_CUDART_ASSERT(hipStreamSynchronize(st));      // the old block goes back to the pool below
        old = nw;
    };
    for (size_t i = 0; i < num_layers; ++i) {
        grow(buffer[i], dim_head, dtype);
        if (scale_dtype_) {
            // one scale per (row, head): (len, heads) / (heads, len) -- as a last dimension of 1
            core::Tensor& sc = scales_[i];
            const size_t old_len = sc.numel() ? sc.size(BSHD ? 0 : 1) : 0;
            if (old_len < new_length) {
                core::Tensor nw = BSHD ? ctx.tensor({new_length, num_heads}, *scale_dtype_) : ctx.tensor({num_heads, new_length}, *scale_dtype_);
                BM_CUDART_ASSERT(hipMemsetAsync(nw.data(), 0, nw.nbytes(), st));
                const size_t esz = core::get_elem_size(*scale_dtype_);
                if (old_len) {
                    if (BSHD) BM_CUDART_ASSERT(hipMemcpyAsync(nw.data(), sc.data(), sc.nbytes(), hipMemcpyDeviceToDevice, st));
                    else BM_CUDART_ASSERT(hipMemcpy2DAsync(nw.data(), new_length * esz, sc.data(), old_len * esz, old_len * esz, num_heads, hipMemcpyDeviceToDevice, st));
                }
                BM_CUDART_ASSERT(hipStreamSynchronize(st));
                sc = nw;
            }
        }
    }
}
// scatter the rows of src (n, heads, dim) into layer `layer` at the buffer rows `placement` names and hand the layer's buffer back
// (attn_encode_group, attention.cpp:513-514: the prompt's keys / values enter the task's buffer here): zl_copy_to_rag_buffer2 with
// one task whose "value" operand is the same tensor
core::Tensor TransformerBuffer::copy(const core::Context& ctx, int layer, const core::Tensor& src, const core::Tensor& placement, int start,
                                     bool need_dequant) {
    check_layer(layer);
    core::Tensor& buf = buffer[layer];
    if (scale_dtype_) {
        // the INT8 cache (transformer_buffer.cu:128-152): the chunk's rows become u8 codes + one fp32 scale per (row, head) at rows
        // start .. start + n - 1 (the reference ignores `placement` here as well); the caller attends over `src` itself, or -- a later
        // chunk, need_dequant -- over the already cached rows brought back to T in front of it
        BM_ASSERT(BSHD && src.ndim() == 3 && *scale_dtype_ == core::DataType::kFloat, "quantised buffers: (len, heads, dim) u8 codes with fp32 scales");
        const int64_t n = (int64_t)src.size(0), len_buf = (int64_t)buf.size(0), row = (int64_t)num_heads * dim_head;
        BM_ASSERT(start >= 0 && start + n <= len_buf, "TransformerBuffer::copy: rows past the buffer");
        const int dt = src.dtype() == core::DataType::kHalf ? ZL_F16 : ZL_BF16;
        zl_stream_t st = (zl_stream_t)ctx.current_cuda_stream();
        core::Tensor& sc = scales_[layer];
        ZL_CK(zl_quant_calc_scale_zp(src.data<uint16_t>(), buf.data<uint8_t>() + (size_t)start * row, sc.data<float>() + (size_t)start * num_heads,
                                     n * (int64_t)num_heads, (int64_t)dim_head, 128, dt, st), "quant_calc_scale");
        if (!(need_dequant && start > 0)) return src;
        core::Tensor out = ctx.tensor({(size_t)len_buf, num_heads, dim_head}, src.dtype());
        BM_CUDART_ASSERT(hipMemsetAsync(out.data(), 0, out.nbytes(), ctx.current_cuda_stream()));
        ZL_CK(zl_dequant_group(buf.data(), sc.data<float>(), out.data<uint16_t>(), (int64_t)start * num_heads, (int64_t)dim_head, 128, dt, st), "dequant_group");
        BM_CUDART_ASSERT(hipMemcpyAsync(out.data<char>() + (size_t)start * row * 2, src.data(), src.nbytes(), hipMemcpyDeviceToDevice, ctx.current_cuda_stream()));
        return out;
    }
    const int64_t n = (int64_t)placement.numel();
    BM_ASSERT(src.numel() == (size_t)n * num_heads * dim_head && placement.dtype() == core::DataType::kInt32, "TransformerBuffer::copy: shape mismatch");
    const int len_buf = (int)buf.size(BSHD ? 0 : 1);
    core::Tensor lens = ctx.tensor_of(std::vector<int>{len_buf});
    core::Tensor table = ctx.tensor_of(std::vector<void*>{buf.data()});
    ZL_CK(zl_copy_to_rag_buffer2(placement.data<int32_t>(), lens.data<int32_t>(), src.data<uint16_t>(), src.data<uint16_t>(),
                                 reinterpret_cast<uint16_t* const*>(table.data()), reinterpret_cast<uint16_t* const*>(table.data()), 1, n, (int64_t)num_heads,
                                 (int64_t)dim_head, BSHD ? 1 : 0, (zl_stream_t)ctx.current_cuda_stream()),
          "copy_to_rag_buffer2");
    BM_CUDART_ASSERT(hipStreamSynchronize(ctx.current_cuda_stream()));      // (lens / table go back to the pool)
    return buf;
}
void copy_to_buffer(int, int, int, int, const core::Tensor*, const core::Tensor&, const core::Tensor&, cudaStream_t, bool) {
    ZL_OFF_PATH("kvcache::copy_to_buffer (the static-batch forward; ragged buffers take copy_to_rag_buffer2)");
}

}  // namespace kvcache


// ---- the paged cache: not on the ragged-buffer path ---------------------------------------------------------------------------
namespace kvcache {
PagedKVCache::PagedKVCache(const PageConfig& pc, int num_layers, int num_heads, int dim_head, core::DataType dtype, bool parallel)
    : KVCache(0, num_layers, num_heads, dim_head, dtype, parallel, false), page_config(pc), block_allocator(pc) {
    ZL_OFF_PATH("kvcache::PagedKVCache (the paged cache; dynamic batching runs on per-task ragged buffers)");
}
PagedKVCache::~PagedKVCache() = default;
const core::Tensor& PagedKVCache::operator[](int) const { ZL_OFF_PATH("kvcache::PagedKVCache"); }
core::Tensor& PagedKVCache::operator[](int) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
core::Tensor& PagedKVCache::key_cache(int) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
core::Tensor& PagedKVCache::value_cache(int) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
const core::Tensor* PagedKVCache::block_table(int) const { ZL_OFF_PATH("kvcache::PagedKVCache"); }
size_t PagedKVCache::add_sequence(const core::Context&, std::vector<int32_t>) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
size_t PagedKVCache::remove_sequence(const core::Context&, size_t) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
size_t PagedKVCache::add_queries(const core::Context&, std::vector<std::vector<int32_t>>) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
void PagedKVCache::resize(const core::Context&, size_t) { ZL_OFF_PATH("kvcache::PagedKVCache"); }
}  // namespace kvcache
