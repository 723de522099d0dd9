// refshim: the handful of cuBLAS / cuBLASLt names the reference's host translation units spell (Int8Linear's IMMA call,
// src/nn/linear/linear.cpp:485-499, 594-620; functions::Gemm::set_compute_type).  Not a BLAS: descriptors are plain
// structs and cublasLtMatmul accepts exactly the product Int8Linear issues -- int8 x int8^T -> int32, alpha 1, beta 0,
// C == D -- and runs it on zl_int8_gemm_nt (the MI355X IMMA-equivalent, bit-exact integer result).  Anything else returns
// CUBLAS_STATUS_NOT_SUPPORTED.  Used only by the compile-the-reference check.
#pragma once
#include <cstdint>
#include <cstring>

#include "cuda_runtime.h"
#include "zhilight_amd.h"

typedef enum { CUBLAS_STATUS_SUCCESS = 0, CUBLAS_STATUS_INVALID_VALUE = 7, CUBLAS_STATUS_NOT_SUPPORTED = 15 } cublasStatus_t;
typedef enum { CUBLAS_OP_N = 0, CUBLAS_OP_T = 1 } cublasOperation_t;
enum { CUBLAS_COMPUTE_16F = 64, CUBLAS_COMPUTE_32F = 68, CUBLAS_COMPUTE_32I = 72 };   // cublasComputeType_t is an int (bm_functions.h)
typedef enum { CUDA_R_16F = 2, CUDA_R_32F = 0, CUDA_R_8I = 3, CUDA_R_32I = 10, CUDA_R_16BF = 14 } cudaDataType_t;
typedef enum { CUBLASLT_EPILOGUE_DEFAULT = 1 } cublasLtEpilogue_t;
typedef enum { CUBLASLT_MATMUL_DESC_TRANSA = 3, CUBLASLT_MATMUL_DESC_TRANSB = 4, CUBLASLT_MATMUL_DESC_EPILOGUE = 7 } cublasLtMatmulDescAttributes_t;
typedef void* cublasHandle_t;
typedef void* cublasLtHandle_t;
struct zl_lt_matmul_desc { int compute, scale; cublasOperation_t transa, transb; cublasLtEpilogue_t epilogue; };
struct zl_lt_layout { cudaDataType_t type; uint64_t rows, cols; int64_t ld; };
typedef zl_lt_matmul_desc* cublasLtMatmulDesc_t;
typedef zl_lt_layout* cublasLtMatrixLayout_t;
typedef struct zl_lt_algo_st cublasLtMatmulAlgo_t;

namespace bmengine {
inline const char* cublasGetErrorString(cublasStatus_t st) {
    return st == CUBLAS_STATUS_SUCCESS ? "success" : st == CUBLAS_STATUS_NOT_SUPPORTED ? "not supported on this boundary" : "invalid value";
}
}  // namespace bmengine
#define BM_CUBLAS_ASSERT(status)                                                                        \
    do {                                                                                                \
        cublasStatus_t v_ = (status);                                                                   \
        if (v_ != CUBLAS_STATUS_SUCCESS)                                                                \
            throw BMEngineException("CUBLAS Error: " #status, __FILE__, __LINE__, __PRETTY_FUNCTION__,  \
                                    bmengine::cublasGetErrorString(v_));                                \
    } while (0)

inline cublasStatus_t cublasLtMatmulDescCreate(cublasLtMatmulDesc_t* d, int compute, cudaDataType_t scale) {
    *d = new zl_lt_matmul_desc{compute, (int)scale, CUBLAS_OP_N, CUBLAS_OP_N, CUBLASLT_EPILOGUE_DEFAULT};
    return CUBLAS_STATUS_SUCCESS;
}
inline cublasStatus_t cublasLtMatmulDescDestroy(cublasLtMatmulDesc_t d) { delete d; return CUBLAS_STATUS_SUCCESS; }
inline cublasStatus_t cublasLtMatmulDescSetAttribute(cublasLtMatmulDesc_t d, cublasLtMatmulDescAttributes_t attr, const void* v, size_t bytes) {
    if (attr == CUBLASLT_MATMUL_DESC_TRANSA && bytes == sizeof(cublasOperation_t)) std::memcpy(&d->transa, v, bytes);
    else if (attr == CUBLASLT_MATMUL_DESC_TRANSB && bytes == sizeof(cublasOperation_t)) std::memcpy(&d->transb, v, bytes);
    else if (attr == CUBLASLT_MATMUL_DESC_EPILOGUE && bytes == sizeof(cublasLtEpilogue_t)) std::memcpy(&d->epilogue, v, bytes);
    else return CUBLAS_STATUS_INVALID_VALUE;
    return CUBLAS_STATUS_SUCCESS;
}
inline cublasStatus_t cublasLtMatrixLayoutCreate(cublasLtMatrixLayout_t* l, cudaDataType_t type, uint64_t rows, uint64_t cols, int64_t ld) {
    *l = new zl_lt_layout{type, rows, cols, ld};
    return CUBLAS_STATUS_SUCCESS;
}
inline cublasStatus_t cublasLtMatrixLayoutDestroy(cublasLtMatrixLayout_t l) { delete l; return CUBLAS_STATUS_SUCCESS; }
// column-major D = alpha * op(A) op(B) + beta * C.  Accepted: A = weight as (K x N, ld K) with op T, B = activations as
// (K x M, ld K), C = D = (N x M, ld N) int32: in row-major words out(M, N) = act(M, K) . weight(N, K)^T.
inline cublasStatus_t cublasLtMatmul(cublasLtHandle_t, cublasLtMatmulDesc_t desc, const void* alpha, const void* A, cublasLtMatrixLayout_t la,
                                     const void* B, cublasLtMatrixLayout_t lb, const void* beta, const void* C, cublasLtMatrixLayout_t lc,
                                     void* D, cublasLtMatrixLayout_t ld, const cublasLtMatmulAlgo_t*, void*, size_t, cudaStream_t stream) {
    const bool ok = desc->compute == CUBLAS_COMPUTE_32I && desc->transa == CUBLAS_OP_T && desc->transb == CUBLAS_OP_N &&
                    la->type == CUDA_R_8I && lb->type == CUDA_R_8I && lc->type == CUDA_R_32I && C == D && lc == ld &&
                    *(const int32_t*)alpha == 1 && *(const int32_t*)beta == 0 && la->rows == lb->rows && (uint64_t)la->ld == la->rows &&
                    (uint64_t)lb->ld == lb->rows && lc->rows == la->cols && lc->cols == lb->cols && (uint64_t)lc->ld == lc->rows;
    if (!ok) return CUBLAS_STATUS_NOT_SUPPORTED;
    const int st = zl_int8_gemm_nt((const int8_t*)B, (const int8_t*)A, (int32_t*)D, (int64_t)lb->cols, (int64_t)la->cols, (int64_t)la->rows, stream);
    return st == 0 ? CUBLAS_STATUS_SUCCESS : CUBLAS_STATUS_INVALID_VALUE;
}
