// overlap_probe.hip -- do two 512-thread / 66 KB-LDS kernels launched on two streams share the CUs, and what does a
// completion-counter hand-off between them cost?  ("alternating streams" plan.)
//
//   producer: 256 workgroups, busy for ~T us, then each bumps a device counter (release) once its work is done
//   consumer: 256 workgroups on ANOTHER stream, launched right behind it; thread 0 polls the counter with a BOUNDED
//             loop (expiry is reported, never a hang), timestamps when it started and when it saw the full count
//
// Reports (100 MHz wall clock): how long after the producer's first workgroup the consumer's first / last workgroup
// started (co-residency), and the time from the LAST producer bump to the consumer workgroups seeing it (hand-off),
// against the same pair launched back to back on one stream.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/overlap_probe tools/ubench/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// method 0: one device counter bumped by every producer workgroup (256 same-address atomics)
// method 1: one flag word per producer workgroup (plain release stores, no atomics); a consumer wave reads all of them
__global__ __launch_bounds__(512, 2) void producer(unsigned long long* ts, int* counter, float* sink, int iters, int method,
                                                   int epoch) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) ts[blockIdx.x * 2] = wall_clock64();
    float a = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
    if (a == 12345.678f) sink[0] = a + lds[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        ts[blockIdx.x * 2 + 1] = wall_clock64();
        if (method == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(counter + 64 + blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(512, 2) void consumer(unsigned long long* ts, int* counter, int expect, int max_polls,
                                                   int* expired, float* sink, int method, int epoch) {
    extern __shared__ unsigned char lds[];
    if (method == 1 && threadIdx.x < 64) {       // wave 0: lane i watches flags i, i + 64, ... of the `expect` producers
        if (threadIdx.x == 0) ts[blockIdx.x * 3] = wall_clock64();
        int polls = 0;
        for (;;) {
            int ok = 1;
            for (int f = threadIdx.x; f < expect; f += 64)
                ok &= __hip_atomic_load(counter + 64 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
            if (__all(ok) || polls >= max_polls) break;
            __builtin_amdgcn_s_sleep(2);
            ++polls;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (threadIdx.x == 0) {
            ts[blockIdx.x * 3 + 1] = wall_clock64();
            ts[blockIdx.x * 3 + 2] = (unsigned long long)polls;
            if (polls >= max_polls) atomicAdd(expired, 1);
        }
    }
    if (method == 0 && threadIdx.x == 0) {
        ts[blockIdx.x * 3] = wall_clock64();
        int polls = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect && polls < max_polls) {
            __builtin_amdgcn_s_sleep(4);
            ++polls;
        }
        ts[blockIdx.x * 3 + 1] = wall_clock64();
        ts[blockIdx.x * 3 + 2] = (unsigned long long)polls;
        if (polls >= max_polls) atomicAdd(expired, 1);
    }
    __syncthreads();
    if (lds[threadIdx.x] == 77 && sink[1] == 3.f) sink[2] = 1.f;
}

int main() {
    const int grid = 256, lds = 66 * 1024, max_polls = 200000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&producer), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&consumer), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    unsigned long long *tp, *tc;
    int *counter, *expired;
    float* sink;
    CK(hipMalloc(&tp, grid * 2 * 8));
    CK(hipMalloc(&tc, grid * 3 * 8));
    CK(hipMalloc(&counter, 4 * (64 + 1024)));
    CK(hipMalloc(&expired, 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(sink, 0, 64));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<unsigned long long> hp(grid * 2), hc(grid * 3);
    int epoch = 0;
    for (int method = 0; method < 2; ++method)
    for (int iters : {2000, 8000}) {
        for (int mode = 0; mode < 3; ++mode) {   // 0: one stream (serial), 1: two streams producer first, 2: two streams consumer first
            double co_first = 0, co_last = 0, hand_min = 0, hand_max = 0, prod_us = 0, total = 0;
            int exp_total = 0, reps = 5;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipMemset(counter, 0, 4 * (64 + 1024)));
                ++epoch;
                CK(hipMemset(expired, 0, 4));
                CK(hipDeviceSynchronize());
                hipStream_t sp = s1, sc = mode == 0 ? s1 : s2;
                if (mode == 2) hipLaunchKernelGGL(consumer, dim3(grid), dim3(512), lds, sc, tc, counter, grid, max_polls, expired, sink, method, epoch);
                hipLaunchKernelGGL(producer, dim3(grid), dim3(512), lds, sp, tp, counter, sink, iters, method, epoch);
                if (mode != 2) hipLaunchKernelGGL(consumer, dim3(grid), dim3(512), lds, sc, tc, counter, grid, max_polls, expired, sink, method, epoch);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hp.data(), tp, grid * 2 * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hc.data(), tc, grid * 3 * 8, hipMemcpyDeviceToHost));
                int ex;
                CK(hipMemcpy(&ex, expired, 4, hipMemcpyDeviceToHost));
                if (r == 0) continue;   // warm-up
                unsigned long long p0 = ~0ull, p1 = 0, pe = 0, c0 = ~0ull, c1 = 0, s0 = ~0ull, s1x = 0;
                for (int b = 0; b < grid; ++b) {
                    p0 = std::min(p0, hp[b * 2]); p1 = std::max(p1, hp[b * 2]); pe = std::max(pe, hp[b * 2 + 1]);
                    c0 = std::min(c0, hc[b * 3]); c1 = std::max(c1, hc[b * 3]);
                    s0 = std::min(s0, hc[b * 3 + 1]); s1x = std::max(s1x, hc[b * 3 + 1]);
                }
                (void)p1;
                co_first += ((double)c0 - (double)p0) / 100.0;
                co_last += ((double)c1 - (double)p0) / 100.0;
                hand_min += ((double)s0 - (double)pe) / 100.0;
                hand_max += ((double)s1x - (double)pe) / 100.0;
                prod_us += ((double)pe - (double)p0) / 100.0;
                total += ((double)s1x - (double)std::min(p0, c0)) / 100.0;
                exp_total += ex;
            }
            const char* names[3] = {"one stream        ", "two streams, P->C ", "two streams, C->P "};
            printf("%s iters %5d  %s producer %6.2f us | consumer start after producer start: first %7.2f last %7.2f us | "
                   "last bump -> seen: first %6.2f last %6.2f us | pair %6.2f us | expired polls %d\n",
                   method ? "flags  " : "counter", iters, names[mode], prod_us / reps, co_first / reps, co_last / reps, hand_min / reps, hand_max / reps,
                   total / reps, exp_total);
        }
    }
    return 0;
}
