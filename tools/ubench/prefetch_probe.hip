// prefetch_probe.hip -- does touching a weight stream's first bytes in the PREVIOUS kernel (which has idle CUs: decode attention
// at batch 1 fills half the chip) make the streaming kernel faster?  A hipGraph chain of pairs (P, S): S = 256 workgroups x 8
// waves streaming r_tiles x 32 KiB each through a 7-deep ring of 1 KiB non-temporal loads (the decode GEMV's shape, no compute);
// P = 256 workgroups x 256 threads touching one dword per LINE bytes of the first pf items of every S wave, either the items of
// the S workgroup with the SAME index (same XCD under round-robin dispatch: the lines land in the L2 that S will ask) or of index
// + 1 (another XCD: only the memory-side Infinity Cache can help).  Buffers rotate over > 256 MB so that nothing is resident
// unless P put it there.  Reported: us per (P, S) pair, P alone, S alone (cold).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/prefetch_probe tools/ubench/prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kT = 512, kRing = 7;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kT, 2) void k_stream(const u4* __restrict__ w, int r_tiles, const unsigned short* vin, unsigned short* vout,
                                                  unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = r_tiles * 4;                         // items per wave
    u4 ring[kRing];
    const unsigned short x = vin[threadIdx.x];
    auto issue = [&](int slot, int j) {
        const size_t it = (size_t)(blockIdx.x * 8 + wave) * n + (j < n ? j : n - 1);
        if (j < n) ring[slot] = __builtin_nontemporal_load(w + it * 64 + lane);
        else ring[slot] = (u4){0, 0, 0, 0};
    };
#pragma unroll
    for (int s = 0; s < kRing; ++s) issue(s, s);
    unsigned acc = x;
    for (int j0 = 0; j0 < n; j0 += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const u4 v = ring[s];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            issue(s, j0 + s + kRing);
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
    if (threadIdx.x < 16) vout[blockIdx.x * 16 + threadIdx.x] = (unsigned short)(x + 1);
}

// P: grid 256 x 256 threads.  pf items per S wave, shift = which S workgroup's items, line = touch stride in bytes (0: nothing)
__global__ __launch_bounds__(256) void k_touch(const unsigned* __restrict__ w, int r_tiles, int pf, int shift, int line,
                                               const unsigned short* vin, unsigned short* vout, unsigned* sink) {
    const unsigned short x = vin[threadIdx.x];
    unsigned acc = x;
    if (line > 0 && pf > 0) {
        const int n = r_tiles * 4, b = (blockIdx.x + shift) % gridDim.x;
        const int per_wave = pf * 1024 / line, total = 8 * per_wave;          // touches per S wave / per S workgroup
        for (int t = threadIdx.x; t < total; t += 256) {
            const int sw = t / per_wave, l = t % per_wave;
            const size_t byte = ((size_t)(b * 8 + sw) * n) * 1024 + (size_t)l * line;
            acc ^= w[byte / 4];
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
    if (threadIdx.x < 16) vout[blockIdx.x * 16 + threadIdx.x] = (unsigned short)(x + 1);
}

int main() {
    const int grid = 256, pairs = 40;
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    unsigned short* vec; CK(hipMalloc(&vec, 8192 * 2)); CK(hipMemset(vec, 0, 8192 * 2));
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    for (int r_tiles : {1, 4, 7}) {
        const size_t wbytes = (size_t)grid * r_tiles * 32 * 1024;
        const int nbuf = (int)((size_t)400 * 1024 * 1024 / wbytes) + 1;
        std::vector<u4*> w(nbuf);
        for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&w[i], wbytes)); CK(hipMemset(w[i], 0x5a, wbytes)); }
        auto run = [&](int mode, int pf, int shift, int line) -> double {   // mode 0: pairs, 1: P only, 2: S only
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
            int flip = 0;
            for (int st = 0; st < pairs; ++st) {
                if (mode != 2) {
                    hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s0, (const unsigned*)w[st % nbuf], r_tiles, pf, shift, line,
                                       vec + flip * 4096, vec + (flip ^ 1) * 4096, sink);
                    flip ^= 1;
                }
                if (mode != 1) {
                    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(kT), 0, s0, w[st % nbuf], r_tiles, vec + flip * 4096, vec + (flip ^ 1) * 4096, sink);
                    flip ^= 1;
                }
            }
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s0));
                for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ge, s0));
                CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            return best * 1e3 / (4 * pairs);
        };
        const double s_only = run(2, 0, 0, 0), p_empty = run(1, 0, 0, 0);
        printf("R=%d (%5.1f MB per stream, %d buffers): S alone %.2f us, empty P alone %.2f us, (empty P, S) %.2f us\n", r_tiles, wbytes / 1e6, nbuf,
               s_only, p_empty, run(0, 0, 0, 0));
        for (int pf : {1, 2, 4, 8, 28}) {
            if (pf > r_tiles * 4 || (pf == 8 && r_tiles == 7)) continue;
            for (int line : {128, 64}) {
                const double p = run(1, pf, 0, line), same = run(0, pf, 0, line), other = run(0, pf, 1, line);
                printf("  pf=%2d items/wave (%5.2f MB) stride %3d: P alone %.2f us | pair same-XCD %.2f -> S %.2f | pair other-XCD %.2f -> S %.2f\n", pf,
                       grid * 8 * pf * 1024 / 1e6, line, p, same, same - p, other, other - p);
            }
        }
        for (int i = 0; i < nbuf; ++i) CK(hipFree(w[i]));
    }
    return 0;
}
