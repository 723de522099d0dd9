// launch_floor.hip -- what one dependent kernel boundary costs under hipGraph replay on this box, by workgroup shape:
// a chain of N launches of (a) an empty kernel, (b) one that makes a dependent global round trip (load -> store),
// (c) the same with a dynamic-LDS request.  build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/launch_floor tools/ubench/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { int v[60]; };
__global__ void k_empty(int* p) { if (p == nullptr) __builtin_trap(); }
__global__ void k_empty_big(Big b, int* p) { if (p == nullptr && b.v[3] == 77) __builtin_trap(); }
__global__ void k_rt(const int* __restrict__ in, int* __restrict__ out) {
    extern __shared__ int sm[];
    const int v = in[threadIdx.x & 63];
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + 1;
}
__global__ void k_rt2(const int* __restrict__ in, int* __restrict__ out) {   // two dependent round trips + barrier
    extern __shared__ int sm[];
    const int v = in[threadIdx.x & 63];
    sm[threadIdx.x] = v;
    __syncthreads();
    const int w = in[(sm[threadIdx.x ^ 1] & 63) + 64];
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + w;
}
template <typename F> float chain(F launch, int n, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (5 * n);
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    int *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 256 * 1024 * 4 * 4); hipMemset(in, 0, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rt), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rt2), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int n = 200;
    Big b{};
    for (int threads : {256, 512}) for (int grid : {192, 256, 512}) {
        printf("grid %4d x %3d thr: empty %.2f us, empty+240B args %.2f, roundtrip %.2f, rt+66KB LDS %.2f, 2 round trips+barrier %.2f, same+66KB %.2f\n", grid, threads,
               chain([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, s, out); }, n, s),
               chain([&] { hipLaunchKernelGGL(k_empty_big, dim3(grid), dim3(threads), 0, s, b, out); }, n, s),
               chain([&] { hipLaunchKernelGGL(k_rt, dim3(grid), dim3(threads), 0, s, in, out); }, n, s),
               chain([&] { hipLaunchKernelGGL(k_rt, dim3(grid), dim3(threads), 66 * 1024, s, in, out); }, n, s),
               chain([&] { hipLaunchKernelGGL(k_rt2, dim3(grid), dim3(threads), 2048, s, in, out); }, n, s),
               chain([&] { hipLaunchKernelGGL(k_rt2, dim3(grid), dim3(threads), 66 * 1024, s, in, out); }, n, s));
    }
    return 0;
}
