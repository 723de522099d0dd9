// bm_functions.h -- the bmengine::functions names the reference's hot-path host code calls, with the reference's
// signatures, on the MI355X C ABI (include/zhilight_amd.h).  What it mirrors (names / signatures; the bodies in
// bm_functions.cpp call zl_* launchers):
//   functions::Gemm                    3rd/bmengine/bmengine/include/bmengine/functions/gemm.h:8-43
//   functions::typecast                .../functions/typecast.h:7
//   functions::Transpose, transpose_2_1  .../functions/transpose.h:8-24
//   functions::concat_tensor, stack_tensor  .../functions/tensor_ops.h:8-15
//   functions::index_select, slice_last_dim .../functions/index_select.h:8-47
//   functions::BinaryElementwiseOp, check_numeric  .../functions/element.h:8-24
//   functions::reduce_abs_max          .../functions/arthmetic.h:10
//   functions::zeros_, ones_, fill     .../functions/init.h:7-12
// Gemm: C = alpha * op(A) op(B) with A (M, K) row-major and B either (N, K) (transB, the layout of every Linear weight)
// or (K, N).  half / bf16 -> zl_gemm_nt (fp32 accumulate, the CUBLAS_COMPUTE_32F setting the reference switches to for
// precision; the 16F accumulate mode has no gfx950 equivalent and is not emulated), int8 -> int32 zl_int8_gemm_nt.
#pragma once
#include "bm_hip.h"
#include "bm_layer.h"

typedef int cublasComputeType_t;     // Gemm::set_compute_type's argument: accepted and ignored (always fp32 / int32 accumulate)

namespace bmengine {
namespace functions {

class Gemm : public core::Layer {
    BM_LAYER_DEF(Gemm)

    Gemm(const core::Context& ctx, core::DataType dtype, bool transA, bool transB, float alpha = 1.0);
    void scale_output(float factor);
    void set_output_type(core::DataType dtype);
    void set_compute_type(cublasComputeType_t compute_type);
    void set_algo_id(int id, int num_search = 20, bool restrict = false);
    void set_A_scale(const core::Tensor& A_scale);
    void set_B_scale(const core::Tensor& B_scale);
    core::Tensor forward(const core::Context& ctx, const core::Tensor& A, const core::Tensor& B, core::Tensor* output = nullptr,
                         const core::Tensor* bias = nullptr);
    core::Tensor batch_3d(const core::Context& ctx, const core::Tensor& A, const core::Tensor& B, core::Tensor* output = nullptr);
};

class Transpose : public core::Layer {
    BM_LAYER_DEF(Transpose)

    Transpose(const core::Context& ctx);
    core::Tensor forward(const core::Context& ctx, const core::Tensor& input, core::Tensor* output = nullptr);   // last two dims
};
core::Tensor transpose_2_1(const core::Context& ctx, const core::Tensor& input, core::Tensor* out_ptr = nullptr);

class BinaryElementwiseOp : public core::Layer {
    BM_LAYER_DEF(BinaryElementwiseOp)

    enum Op { Add, Sub, Mul, Div, Max };
    BinaryElementwiseOp(const core::Context& ctx, Op op);
    core::Tensor forward(const core::Context& ctx, const core::Tensor& x, const core::Tensor& y, core::Tensor* out = nullptr);
    void inplace(const core::Context& ctx, const core::Tensor& x, const core::Tensor& y);
    core::Tensor broadcast_y(const core::Context& ctx, const core::Tensor& x, const core::Tensor& y);   // y: one value per row of x
};

void check_numeric(const core::Context& ctx, const core::Tensor& tensor);     // throws on NaN / Inf
core::Tensor typecast(const core::Context& ctx, const core::Tensor& in, core::DataType out_type);
core::Tensor concat_tensor(const core::Context& ctx, const core::Tensor& A, const core::Tensor& B, int dim = -1);
core::Tensor concat_tensor(const core::Context& ctx, const std::vector<core::Tensor>& tensors, int dim = 0);
core::Tensor stack_tensor(const core::Context& ctx, const std::vector<core::Tensor>& tensors);
core::Tensor index_select(const core::Context& ctx, const core::Tensor& input, int dim, const core::Tensor& index,
                          core::Tensor* out = nullptr);
core::Tensor slice_last_dim(const core::Context& ctx, const core::Tensor& tensor, int from, int len, core::Tensor* out_ptr = nullptr);
// index_select.h:32-39: output[..., i] = input[..., from + i] for i < output.size(-1); columns past the input's width are zero
// when padding_zero (otherwise `to` = from + output.size(-1) must lie inside the input)
void copy_last_dim(hipStream_t stream, const core::Tensor& input, core::Tensor& output, int from, int to = -1, bool padding_zero = false);
// tensor_ops.h:13-14: C[l, m, :] = [A[l, m, :] | B[l, :]] for A (L, M, a), B (L, b)
core::Tensor concat_broadcast_b(const core::Context& ctx, const core::Tensor& A, const core::Tensor& B);
core::Tensor reduce_abs_max(const core::Context& ctx, const core::Tensor& a, int dim = 0);
// the MoE dispatch route's index plumbing (init.h:10, element.h:25, scatter.h:7-13, sort.h:8-17): int32 forms
core::Tensor arange(const core::Context& ctx, int start, int end, int step);
core::Tensor divide(const core::Context& ctx, const core::Tensor& a, float divisor);
void scatter_update_dim0(const core::Context& ctx, core::Tensor& dst, const core::Tensor& dst_index, const core::Tensor& src, const core::Tensor& src_index);
std::pair<core::Tensor, core::Tensor> sort_pair_1d(const core::Context& ctx, const core::Tensor& keys, const core::Tensor& values, int max_key);
std::pair<core::Tensor, core::Tensor> sort_with_indices_1d(const core::Context& ctx, const core::Tensor& keys, int max_key);
void zeros_(const core::Context& ctx, const core::Tensor& x);
void ones_(const core::Context& ctx, const core::Tensor& x);
void fill(const core::Context& ctx, const core::Tensor& x, float value);

// What the batch generator runs on logits between decode steps (softmax.h:8-12, topk.h:9-23); defined in host_generator_ext.cpp (part of the
// host library, not of zl_internals) over zl_softmax_rows / zl_topk_rows
void softmax(const core::Context& ctx, const core::Tensor& logits, const core::Tensor& output, float temperature = 1.0f);   // rows of the last dimension
void bitonic_topk(const core::Context& ctx, const core::Tensor& x, const core::Tensor& out, const core::Tensor& pos);
class TopK : public core::Layer {      // forward: (values (batch, top) descending, int32 positions (batch, top)) of a (batch, n) input
    BM_LAYER_DEF(TopK)
    explicit TopK(const core::Context& ctx);
    std::pair<core::Tensor, core::Tensor> forward(const core::Context& ctx, const core::Tensor& inp, int top);
};

}  // namespace functions
}  // namespace bmengine
