// bm_functions.cpp -- bmengine::functions on the MI355X C ABI (see bm_functions.h for what each name mirrors).
#include "bm_functions.h"

#include <cstring>

#include "zhilight_amd.h"

namespace bmengine {
namespace functions {

using core::Context;
using core::DataType;
using core::Tensor;

namespace {
void zl_check(int st, const char* what) {
    if (st != ZL_OK) throw BMEngineException(std::string(what) + ": " + zl_status_string(st), __FILE__, __LINE__, __func__);
}
zl_stream_t st_of(const Context& ctx) { return (zl_stream_t)ctx.current_cuda_stream(); }
int zdt(DataType t) {   // the 16-bit activation code of the C ABI
    BM_ASSERT(t == DataType::kHalf || t == DataType::kBFloat16, std::string("half / bfloat16 expected, got ") + get_data_type_name(t));
    return t == DataType::kHalf ? ZL_F16 : ZL_BF16;
}
int elem_code(DataType t) {   // zl_elem_t shares bmengine's enumerator values
    BM_ASSERT((int)t <= (int)DataType::kBFloat16, std::string("element type not supported here: ") + get_data_type_name(t));
    return (int)t;
}
size_t rows_of(const Tensor& t) { return t.numel() / t.size(-1); }
Tensor contiguous(const Context& ctx, const Tensor& t) {
    if (t.is_continuous()) return t;
    BM_ASSERT_EQ(t.ndim(), 2, "only 2-d strided tensors can be compacted");
    Tensor out = ctx.tensor(t.shape(), t.dtype());
    const size_t es = core::get_elem_size(t.dtype());
    BM_ASSERT_EQ(t.stride(1), (size_t)1, "the last dimension must be dense");
    zl_check(zl_copy_2d(t.data(), t.stride(0) * es, out.data(), t.size(1) * es, t.size(1) * es, t.size(0), st_of(ctx)), "copy_2d");
    return out;
}
}  // namespace

// ---- Gemm ------------------------------------------------------------------------------------------------------------
class Gemm::impl {
public:
    DataType dtype, out_type;
    bool transA, transB;
    float alpha;
    const void* a_scale = nullptr;
    const void* b_scale = nullptr;
};
Gemm::Gemm(const Context&, DataType dtype, bool transA, bool transB, float alpha) : pimpl(new impl) {
    pimpl->dtype = dtype;
    pimpl->out_type = dtype == DataType::kInt8 ? DataType::kInt32 : dtype == DataType::kFP8_E4M3 ? DataType::kHalf : dtype;
    pimpl->transA = transA;
    pimpl->transB = transB;
    pimpl->alpha = alpha;
    if (dtype == DataType::kInt8 || dtype == DataType::kFP8_E4M3) BM_ASSERT(!transA && transB, "int8 / fp8 products need NT");
}
Gemm::~Gemm() = default;
void Gemm::scale_output(float factor) { pimpl->alpha *= factor; }
void Gemm::set_output_type(DataType dtype) {
    // fp32 output of a half / bf16 product (the MoE router's logits): zl_gemm_nt_f32, no rounding to T in between
    if (dtype == DataType::kFloat && (pimpl->dtype == DataType::kHalf || pimpl->dtype == DataType::kBFloat16)) {
        pimpl->out_type = DataType::kFloat;
        return;
    }
    BM_ASSERT(dtype == pimpl->out_type, "Gemm::set_output_type: this boundary produces the input type (int32 for int8, half for fp8; fp32 on request for half / bf16)");
}
void Gemm::set_compute_type(cublasComputeType_t) {}    // always fp32 (int32) accumulation
void Gemm::set_algo_id(int, int, bool) {}
void Gemm::set_A_scale(const Tensor& s) { pimpl->a_scale = s.data(); }
void Gemm::set_B_scale(const Tensor& s) { pimpl->b_scale = s.data(); }

Tensor Gemm::forward(const Context& ctx, const Tensor& A0, const Tensor& B0, Tensor* output, const Tensor* bias) {
    // (batch, M, K) x (batch, ., .): the reference's forward is cublasGemmStridedBatched for these (gemm.cpp:505-542; MLAImpl's absorbed
    // projections call it that way, multi_head_latent_attention.cpp:1033, 1087-1091) -- one product per batch entry here
    if (A0.ndim() == 3 && B0.ndim() == 3 && !bias && A0.size(0) == B0.size(0) && A0.is_continuous() && B0.is_continuous())
        return batch_3d(ctx, A0, B0, output);
    BM_ASSERT(A0.ndim() >= 2 && B0.ndim() == 2, "Gemm: A (..., M, K) x B 2-d, or two dense 3-d operands with equal batch");
    BM_ASSERT_EQ(A0.dtype(), pimpl->dtype, "Gemm: A dtype");
    BM_ASSERT_EQ(B0.dtype(), pimpl->dtype, "Gemm: B dtype");
    Transpose tr(ctx);
    // bring both operands to the NT form the kernels stream: A (M, K) rows, B (N, K) rows
    Tensor A = pimpl->transA ? tr.forward(ctx, A0) : A0;
    Tensor B = pimpl->transB ? contiguous(ctx, B0) : tr.forward(ctx, B0);
    const int64_t k = A.size(-1), n = B.size(0), m = rows_of(A);
    BM_ASSERT_EQ((int64_t)B.size(1), k, "Matrix dimensions mismatch");
    if (A.ndim() > 2) BM_ASSERT(A.is_continuous(), "Gemm: a batched A must be dense");
    const int64_t lda = A.ndim() == 2 && m > 1 ? (int64_t)A.stride(0) : k;      // (the stride of a single row is meaningless: views leave 1 there)
    BM_ASSERT_EQ(A.stride(-1), (size_t)1, "Gemm: the last dimension of A must be dense");
    std::vector<size_t> oshape = A.shape();
    oshape.back() = n;
    if (output && !output->is_continuous()) {
        // a strided (rows, n) output -- MLAImpl's data-parallel decode writes the absorbed value projection of each head straight into
        // a (rows, heads, v_head_dim) tensor viewed head-major (multi_head_latent_attention.cpp:1203-1212): the product into a dense
        // block, then a pitched copy of its rows
        BM_ASSERT(output->ndim() == 2 && output->stride(1) == 1 && (int64_t)output->size(0) == m && (int64_t)output->size(1) == n,
                  "Gemm: a strided output is (rows, n) with a dense last dimension");
        BM_ASSERT_EQ(output->dtype(), pimpl->out_type, "Gemm: output dtype");
        Tensor dense = ctx.tensor({(size_t)m, (size_t)n}, pimpl->out_type);
        forward(ctx, A0, B0, &dense, bias);
        const size_t row_bytes = (size_t)n * core::get_elem_size(pimpl->out_type);
        zl_check(zl_copy_2d(dense.data(), row_bytes, output->data(), m > 1 ? output->stride_bytes(0) : row_bytes, row_bytes, m, st_of(ctx)),
                 "Gemm (strided output rows)");
        return *output;
    }
    Tensor out = output ? *output : ctx.tensor(oshape, pimpl->out_type);
    BM_ASSERT_EQ(out.numel(), (size_t)(m * n), "Gemm: output shape mismatch");
    BM_ASSERT_EQ(out.dtype(), pimpl->out_type, "Gemm: output dtype");
    BM_ASSERT(out.is_continuous(), "Gemm: output must be dense");
    if (pimpl->dtype == DataType::kInt8) {
        BM_ASSERT(!bias, "Gemm(int8): no bias");
        BM_ASSERT_EQ(lda, k, "Gemm(int8): A must be dense");
        zl_check(zl_int8_gemm_nt(A.data<int8_t>(), B.data<int8_t>(), out.data<int32_t>(), m, n, k, st_of(ctx)), "Gemm(int8)");
        return out;
    }
    if (pimpl->dtype == DataType::kFP8_E4M3) {
        BM_ASSERT(!bias && pimpl->a_scale && pimpl->b_scale, "Gemm(fp8): per-tensor scales (set_A_scale / set_B_scale), no bias");
        BM_ASSERT_EQ(lda, k, "Gemm(fp8): A must be dense");
        zl_check(zl_fp8_gemm_nt(A.data<uint8_t>(), B.data<uint8_t>(), (const float*)pimpl->a_scale, (const float*)pimpl->b_scale,
                                out.data<uint16_t>(), m, n, k, st_of(ctx)), "Gemm(fp8)");
        return out;
    }
    const int dt = zdt(pimpl->dtype);
    if (pimpl->out_type == DataType::kFloat) {
        BM_ASSERT(!bias || !bias->numel(), "Gemm(fp32 output): no bias");
        zl_check(zl_gemm_nt_f32(A.data<uint16_t>(), lda, B.data<uint16_t>(), out.data<float>(), m, n, k, pimpl->alpha, dt, st_of(ctx)), "Gemm (fp32 output)");
        return out;
    }
    const uint16_t* bp = bias && bias->numel() ? bias->data<uint16_t>() : nullptr;
    if (m <= 4 || k % 128 != 0)
        zl_check(zl_gemm_nt_small_m(A.data<uint16_t>(), lda, B.data<uint16_t>(), bp, out.data<uint16_t>(), m, n, k, pimpl->alpha, dt, nullptr,
                                    0.f, st_of(ctx)), "Gemm (row-streaming)");
    else
        zl_check(zl_gemm_nt(A.data<uint16_t>(), lda, B.data<uint16_t>(), bp, out.data<uint16_t>(), m, n, k, pimpl->alpha, dt, st_of(ctx)),
                 "Gemm");
    return out;
}
Tensor Gemm::batch_3d(const Context& ctx, const Tensor& A, const Tensor& B, Tensor* output) {
    BM_ASSERT(A.ndim() == 3 && B.ndim() == 3 && A.size(0) == B.size(0), "batch_3d: (B, M, K) x (B, N, K) or (B, K, N)");
    const size_t m = pimpl->transA ? A.size(2) : A.size(1), n = pimpl->transB ? B.size(1) : B.size(2);
    if (output && !output->is_continuous()) {                  // (batch, m, n) with arbitrary batch / row strides: entry by entry
        BM_ASSERT(output->ndim() == 3 && output->size(0) == A.size(0) && output->size(1) == m && output->size(2) == n, "batch_3d: output shape");
        for (size_t b = 0; b < A.size(0); ++b) {
            Tensor o = output->index_dim0(b);
            forward(ctx, A.index_dim0(b), B.index_dim0(b), &o, nullptr);
        }
        return *output;
    }
    Tensor out = output ? output->view({A.size(0), m, n}) : ctx.tensor({A.size(0), m, n}, pimpl->out_type);
    for (size_t b = 0; b < A.size(0); ++b) {
        Tensor o = out.index_dim0(b);
        forward(ctx, A.index_dim0(b), B.index_dim0(b), &o, nullptr);
    }
    return output ? *output : out;
}

// ---- the MoE dispatch route's index plumbing (init.h:10, element.h:25, scatter.h:7-13, sort.h:8-17) ----------------------------------
Tensor arange(const Context& ctx, int start, int end, int step) {
    BM_ASSERT(step != 0 && (end - start) / step >= 0, "arange: empty or reversed range");
    const size_t num = size_t(end - start) / step;
    Tensor out = ctx.tensor({num}, DataType::kInt32);
    if (num) zl_check(zl_arange_i32(out.data<int32_t>(), start, step, num, st_of(ctx)), "arange");
    return out;
}
Tensor divide(const Context& ctx, const Tensor& a, float divisor) {
    // the reference divides in the element type (UnaryOpDivide<T>: T(divisor)); the dispatch route calls it on int32 token indices
    BM_ASSERT(a.dtype() == DataType::kInt32 && a.is_continuous(), "divide: int32 (the index form) is what this boundary provides");
    BM_ASSERT((float)(int)divisor == divisor && (int)divisor != 0, "divide: integral divisor");
    Tensor out = ctx.tensor(a.shape(), a.dtype());
    if (a.numel()) zl_check(zl_divide_i32(a.data<int32_t>(), out.data<int32_t>(), (int)divisor, a.numel(), st_of(ctx)), "divide");
    return out;
}
void scatter_update_dim0(const Context& ctx, Tensor& dst, const Tensor& dst_index, const Tensor& src, const Tensor& src_index) {
    BM_ASSERT_EQ(dst.dtype(), src.dtype(), "src dst dtype mismatch");
    BM_ASSERT(dst.ndim() == 2 && src.ndim() == 2 && dst.size(-1) == src.size(-1), "scatter_update_dim0: (X, D) <- (Y, D)");
    BM_ASSERT(dst.is_continuous() && src.is_continuous(), "scatter_update_dim0: dense operands");
    BM_ASSERT(dst_index.dtype() == DataType::kInt32 && dst_index.ndim() == 1, "dst_index is not 1-d int");
    const bool has_src = src_index.numel() > 0;
    if (has_src) BM_ASSERT(src_index.dtype() == DataType::kInt32 && src_index.numel() == dst_index.numel(), "src_index / dst_index mismatch");
    if (!dst_index.numel()) return;
    const size_t row_bytes = src.size(-1) * core::get_elem_size(src.dtype());
    BM_ASSERT(row_bytes % 2 == 0, "scatter_update_dim0: rows of an even number of bytes");
    zl_check(zl_scatter_update_dim0(dst.data(), dst_index.data<int32_t>(), src.data(), has_src ? src_index.data<int32_t>() : nullptr, dst_index.numel(),
                                    row_bytes, dst.size(0), src.size(0), st_of(ctx)), "scatter_update_dim0");
}
std::pair<Tensor, Tensor> sort_pair_1d(const Context& ctx, const Tensor& keys, const Tensor& values, int max_key) {
    BM_ASSERT(keys.ndim() == 1 && values.ndim() == 1 && keys.numel() == values.numel(), "sort_pair_1d: two 1-d tensors of one length");
    BM_ASSERT(keys.dtype() == DataType::kInt32 && values.dtype() == DataType::kInt32, "sort_pair_1d: int32 keys (non-negative) and values");
    Tensor ko = ctx.tensor(keys.shape(), keys.dtype()), vo = ctx.tensor(values.shape(), values.dtype());
    if (!keys.numel()) return {ko, vo};
    Tensor wsp = ctx.tensor({keys.numel() * 2}, DataType::kInt32);
    zl_check(zl_sort_pairs_i32(keys.data<int32_t>(), values.data<int32_t>(), ko.data<int32_t>(), vo.data<int32_t>(), wsp.data(), keys.numel(), max_key,
                               st_of(ctx)), "sort_pair_1d");
    return {ko, vo};
}
std::pair<Tensor, Tensor> sort_with_indices_1d(const Context& ctx, const Tensor& keys, int max_key) {
    return sort_pair_1d(ctx, keys, arange(ctx, 0, (int)keys.numel(), 1), max_key);
}

// ---- Transpose ---------------------------------------------------------------------------------------------------------
class Transpose::impl {};
Transpose::Transpose(const Context&) : pimpl(new impl) {}
Transpose::~Transpose() = default;
Tensor Transpose::forward(const Context& ctx, const Tensor& input, Tensor* output) {
    BM_ASSERT(input.ndim() == 2 || input.ndim() == 3, "Transpose: 2-d, or 3-d (batch, rows, cols)");
    const Tensor in = input.ndim() == 2 ? contiguous(ctx, input) : input;
    BM_ASSERT(in.is_continuous(), "Transpose: dense input");
    std::vector<size_t> shape = in.shape();
    std::swap(shape[shape.size() - 1], shape[shape.size() - 2]);
    Tensor out = output ? *output : ctx.tensor(shape, in.dtype(), in.name());
    BM_ASSERT_EQ(out.numel(), in.numel(), "Transpose: output shape mismatch");
    const size_t rows = in.size(-2), cols = in.size(-1), es = core::get_elem_size(in.dtype()), batch = in.numel() / (rows * cols);
    BM_ASSERT(es == 1 || es == 2 || es == 4, "Transpose: 1, 2 or 4-byte elements");
    for (size_t b = 0; b < batch; ++b)
        zl_check(zl_transpose_2d((const char*)in.data() + b * rows * cols * es, (char*)out.data() + b * rows * cols * es, rows, cols, (int)es,
                                 st_of(ctx)), "Transpose");
    out.quant_scale = input.quant_scale;
    return out;
}
Tensor transpose_2_1(const Context& ctx, const Tensor& input, Tensor* out_ptr) {
    // (batch?, d1, d2, last) -> (batch?, d2, d1, last): a transpose of (d1, d2) whose element is a run of `last` values
    BM_ASSERT(input.ndim() == 3 || input.ndim() == 4, "transpose_2_1: 3-d or 4-d");
    BM_ASSERT(input.is_continuous(), "transpose_2_1: dense input");
    std::vector<size_t> shape = input.shape();
    const size_t nd = shape.size(), d1 = shape[nd - 3], d2 = shape[nd - 2], last = shape[nd - 1];
    std::swap(shape[nd - 3], shape[nd - 2]);
    Tensor out = out_ptr ? *out_ptr : ctx.tensor(shape, input.dtype());
    const size_t es = core::get_elem_size(input.dtype()), run = last * es, batch = nd == 4 ? shape[0] : 1;
    // row j of the output's (d2, d1) grid gathers the d1 runs spaced d2 * run apart: one pitched copy per j
    for (size_t b = 0; b < batch; ++b)
        for (size_t j = 0; j < d2; ++j)
            zl_check(zl_copy_2d((const char*)input.data() + (b * d1 * d2 + j) * run, d2 * run, (char*)out.data() + (b * d1 * d2 + j * d1) * run,
                                run, run, d1, st_of(ctx)), "transpose_2_1");
    return out;
}

// ---- element-wise ------------------------------------------------------------------------------------------------------
class BinaryElementwiseOp::impl {
public:
    Op op;
};
BinaryElementwiseOp::BinaryElementwiseOp(const Context&, Op op) : pimpl(new impl) { pimpl->op = op; }
BinaryElementwiseOp::~BinaryElementwiseOp() = default;
Tensor BinaryElementwiseOp::forward(const Context& ctx, const Tensor& x, const Tensor& y, Tensor* out) {
    BM_ASSERT_EQ(x.numel(), y.numel(), "BinaryElementwiseOp: size mismatch");
    BM_ASSERT_EQ(x.dtype(), y.dtype(), "BinaryElementwiseOp: dtype mismatch");
    Tensor ret = out ? *out : ctx.tensor(x.size(), x.dtype());
    zl_check(zl_binary_op(x.data(), y.data(), ret.data(), 1, x.numel(), (int)pimpl->op, 0, elem_code(x.dtype()), st_of(ctx)),
             "BinaryElementwiseOp");
    return ret;
}
void BinaryElementwiseOp::inplace(const Context& ctx, const Tensor& x, const Tensor& y) {
    Tensor alias = x;
    forward(ctx, x, y, &alias);
}
Tensor BinaryElementwiseOp::broadcast_y(const Context& ctx, const Tensor& x, const Tensor& y) {
    BM_ASSERT_EQ(x.dtype(), y.dtype(), "dtype mismatch");
    BM_ASSERT(x.ndim() > 1, "wrong dim");
    Tensor ret = ctx.tensor(x.size(), x.dtype());
    if (y.size(-1) == 1) {   // one value per row of x
        BM_ASSERT_EQ(x.ndim(), y.ndim(), "wrong dim");
        BM_ASSERT_EQ(y.numel(), rows_of(x), "shape mismatch");
        zl_check(zl_binary_op(x.data(), y.data(), ret.data(), rows_of(x), x.size(-1), (int)pimpl->op, 1, elem_code(x.dtype()), st_of(ctx)),
                 "broadcast_y");
        return ret;
    }
    BM_ASSERT(y.ndim() + 1 <= x.ndim() && x.numel() % y.numel() == 0, "wrong dim");
    for (int i = 1; i <= y.ndim(); ++i) BM_ASSERT_EQ(x.size(-i), y.size(-i), "shape mismatch of dim:-" + std::to_string(i));
    zl_check(zl_binary_op(x.data(), y.data(), ret.data(), x.numel() / y.numel(), y.numel(), (int)pimpl->op, 2, elem_code(x.dtype()),
                          st_of(ctx)), "broadcast_y");
    return ret;
}

void check_numeric(const Context& ctx, const Tensor& tensor) {
    if (tensor.numel() == 0) return;
    Tensor counter = ctx.tensor({1}, DataType::kInt32);
    BM_HIPRT_ASSERT(hipMemsetAsync(counter.data(), 0, 4, ctx.current_cuda_stream()));
    zl_check(zl_count_nonfinite(tensor.data(), tensor.numel(), elem_code(tensor.dtype()), counter.data<int32_t>(), st_of(ctx)),
             "check_numeric");
    int32_t bad = 0;
    counter.to_buffer(&bad, ctx.current_cuda_stream());
    BM_ASSERT(bad == 0, "check_numeric: " + std::to_string(bad) + " non-finite values in " + tensor.info());
}

Tensor typecast(const Context& ctx, const Tensor& in, DataType out_type) {
    if (in.dtype() == out_type) return in;
    BM_ASSERT(in.is_continuous(), "typecast: dense input");
    Tensor out = ctx.tensor(in.shape(), out_type, in.name());
    zl_check(zl_cast(in.data(), elem_code(in.dtype()), out.data(), elem_code(out_type), in.numel(), st_of(ctx)), "typecast");
    return out;
}

// ---- concat / slice / gather ---------------------------------------------------------------------------------------------
Tensor concat_tensor(const Context& ctx, const std::vector<Tensor>& tensors, int dim) {
    BM_ASSERT(!tensors.empty(), "concat_tensor: nothing to concatenate");
    const Tensor& first = tensors[0];
    const int d = first.normalize_dim(dim);
    std::vector<size_t> shape = first.shape();
    shape[d] = 0;
    for (const Tensor& t : tensors) {
        BM_ASSERT_EQ(t.ndim(), first.ndim(), "concat_tensor: rank mismatch");
        BM_ASSERT_EQ(t.dtype(), first.dtype(), "concat_tensor: dtype mismatch");
        BM_ASSERT(t.is_continuous(), "concat_tensor: dense inputs");
        for (int i = 0; i < first.ndim(); ++i)
            if (i != d) BM_ASSERT_EQ(t.size(i), first.size(i), "concat_tensor: shape mismatch");
        shape[d] += t.size(d);
    }
    Tensor out = ctx.tensor(shape, first.dtype());
    size_t outer = 1, inner = core::get_elem_size(first.dtype());
    for (int i = 0; i < d; ++i) outer *= shape[i];
    for (int i = d + 1; i < first.ndim(); ++i) inner *= shape[i];
    const size_t out_pitch = shape[d] * inner;
    size_t off = 0;
    for (const Tensor& t : tensors) {
        const size_t w = t.size(d) * inner;
        if (w == 0) continue;
        zl_check(zl_copy_2d(t.data(), w, (char*)out.data() + off, out_pitch, w, outer, st_of(ctx)), "concat_tensor");
        off += w;
    }
    return out;
}
Tensor concat_tensor(const Context& ctx, const Tensor& A, const Tensor& B, int dim) {
    if (A.numel() == 0) return B;
    if (B.numel() == 0) return A;
    return concat_tensor(ctx, std::vector<Tensor>{A, B}, dim);
}
Tensor stack_tensor(const Context& ctx, const std::vector<Tensor>& tensors) {
    BM_ASSERT(!tensors.empty(), "stack_tensor: nothing to stack");
    std::vector<Tensor> lifted;
    for (const Tensor& t : tensors) {
        std::vector<size_t> s = t.shape();
        s.insert(s.begin(), 1);
        lifted.push_back(t.view(s));
    }
    return concat_tensor(ctx, lifted, 0);
}
Tensor slice_last_dim(const Context& ctx, const Tensor& tensor, int from, int len, Tensor* out_ptr) {
    BM_ASSERT(from >= 0 && len > 0 && (size_t)(from + len) <= tensor.size(-1), "slice_last_dim out of range");
    BM_ASSERT(tensor.is_continuous(), "slice_last_dim: dense input");
    std::vector<size_t> shape = tensor.shape();
    shape.back() = len;
    Tensor out = out_ptr ? *out_ptr : ctx.tensor(shape, tensor.dtype());
    const size_t es = core::get_elem_size(tensor.dtype());
    zl_check(zl_copy_2d((const char*)tensor.data() + from * es, tensor.size(-1) * es, out.data(), len * es, len * es, rows_of(tensor),
                        st_of(ctx)), "slice_last_dim");
    return out;
}
void copy_last_dim(hipStream_t stream, const Tensor& input, Tensor& output, int from, int to, bool padding_zero) {
    BM_ASSERT_EQ(input.dtype(), output.dtype(), "type mismatch");
    BM_ASSERT_EQ(input.ndim(), output.ndim(), "rank mismatch");
    BM_ASSERT(input.is_continuous() && output.is_continuous() && from >= 0, "copy_last_dim: dense tensors");
    const size_t width = output.size(-1), in_w = input.size(-1), rows = output.numel() / width;
    BM_ASSERT_EQ(rows, input.numel() / in_w, "copy_last_dim: row count mismatch");
    if (to == -1) to = from + (int)width;
    if (!padding_zero) BM_ASSERT_LE((size_t)to, in_w, "to out of range");
    const size_t es = core::get_elem_size(input.dtype());
    const size_t take = (size_t)from >= in_w ? 0 : std::min(width, in_w - from);      // columns that exist in the input
    if (take < width) BM_HIPRT_ASSERT(hipMemsetAsync(output.data(), 0, output.nbytes(), stream));
    if (take) zl_check(zl_copy_2d((const char*)input.data() + from * es, in_w * es, output.data(), width * es, take * es, rows, (zl_stream_t)stream),
                       "copy_last_dim");
}
Tensor concat_broadcast_b(const Context& ctx, const Tensor& A, const Tensor& B) {
    BM_ASSERT_EQ(A.ndim(), 3, "");
    BM_ASSERT_EQ(B.ndim(), 2, "");
    BM_ASSERT_EQ(A.size(0), B.size(0), "");
    BM_ASSERT_EQ(A.dtype(), B.dtype(), "type mismatch");
    BM_ASSERT(A.is_continuous() && B.is_continuous(), "concat_broadcast_b: dense tensors");
    const size_t L = A.size(0), M = A.size(1), pa = A.size(2), pb = B.size(1), pc = pa + pb, es = core::get_elem_size(A.dtype());
    Tensor out = ctx.tensor({L, M, pc}, A.dtype());
    zl_check(zl_copy_2d(A.data(), pa * es, out.data(), pc * es, pa * es, L * M, st_of(ctx)), "concat_broadcast_b");
    // B's row l under every m: out viewed as (L, M * pc), the row's pb columns at offset m * pc + pa -- one strided copy per m
    for (size_t m = 0; m < M; ++m)
        zl_check(zl_copy_2d(B.data(), pb * es, (char*)out.data() + (m * pc + pa) * es, M * pc * es, pb * es, L, st_of(ctx)), "concat_broadcast_b");
    return out;
}
Tensor index_select(const Context& ctx, const Tensor& input, int dim, const Tensor& index, Tensor* out) {
    BM_ASSERT_EQ(index.dtype(), DataType::kInt32, "index_select: int32 index");
    BM_ASSERT_EQ(index.ndim(), 1, "index_select: 1-d index");
    BM_ASSERT(input.is_continuous(), "index_select: dense input");
    const int d = input.normalize_dim(dim);
    std::vector<size_t> shape = input.shape();
    size_t outer = 1, inner = core::get_elem_size(input.dtype());
    for (int i = 0; i < d; ++i) outer *= shape[i];
    for (int i = d + 1; i < input.ndim(); ++i) inner *= shape[i];
    shape[d] = index.numel();
    Tensor ret = out ? *out : ctx.tensor(shape, input.dtype());
    BM_ASSERT_EQ(ret.numel(), core::get_numel(shape), "index_select: output shape mismatch");
    zl_check(zl_index_select(input.data(), ret.data(), index.data<int32_t>(), outer, input.size(d), index.numel(), inner, st_of(ctx)),
             "index_select");
    return ret;
}
Tensor reduce_abs_max(const Context& ctx, const Tensor& a, int dim) {
    BM_ASSERT_EQ(a.ndim(), 2, "not 2-D tensor");
    const Tensor v = a.normalize_dim(dim) == 0 ? Transpose(ctx).forward(ctx, a) : contiguous(ctx, a);
    Tensor out = ctx.tensor({v.size(0)}, v.dtype());
    zl_check(zl_reduce_abs_max(v.data(), out.data(), v.size(0), v.size(1), elem_code(v.dtype()), st_of(ctx)), "reduce_abs_max");
    return out;
}

// ---- init ----------------------------------------------------------------------------------------------------------------
void zeros_(const Context& ctx, const Tensor& x) {
    if (x.numel()) BM_HIPRT_ASSERT(hipMemsetAsync(x.data(), 0, x.nbytes(), ctx.current_cuda_stream()));
}
void fill(const Context& ctx, const Tensor& x, float value) {
    if (!x.numel()) return;
    if (value == 0.f) return zeros_(ctx, x);
    // value = 0 * x + value, through the broadcast path: a one-element operand of x's type
    Tensor one = ctx.tensor({1}, DataType::kFloat);
    one.from_buffer(&value, false, ctx.current_cuda_stream());
    Tensor v = typecast(ctx, one, x.dtype());
    zeros_(ctx, x);
    zl_check(zl_binary_op(x.data(), v.data(), x.data(), 1, x.numel(), 0, 1, elem_code(x.dtype()), st_of(ctx)), "fill");
}
void ones_(const Context& ctx, const Tensor& x) { fill(ctx, x, 1.f); }

}  // namespace functions
}  // namespace bmengine
