// stream_probe.hip -- what does the SHAPE of a weight stream cost?  A chain of dependent kernels (hipGraph, 256 workgroups
// x 8 waves, ring of 7 x 1 KiB non-temporal loads per wave, no compute), bytes per workgroup as in the batch-1 decode
// GEMVs, with the item order and the side stream varied:
//   order 0: a wave reads one contiguous run of items            order 1: the phase kernel's order -- in phase ph wave w
//            reads item 8 ph + w of each of the workgroup's R tiles (32 items per tile)
//   meta 0: nibbles only     meta 1: + one 64-byte load per item from a second array (ZLW4M's scale|zero words)
//   meta 2: the 64 bytes ride at the end of the item (1088-byte items, one stream)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_probe tools/ubench/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kT = 512, kRing = 7;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int ORDER, int META>
__global__ __launch_bounds__(kT, 2) void k_stream(const u4* __restrict__ w, const unsigned* __restrict__ meta, int r_tiles,
                                                  const unsigned short* vin, unsigned short* vout, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = r_tiles * 4;                         // items per wave (32 items per tile / 8 waves)
    const size_t istride = META == 2 ? 68 : 64;        // u4 per item
    auto item_of = [&](int j) -> size_t {
        if (ORDER == 0) return (size_t)(blockIdx.x * 8 + wave) * n + j;
        const int ph = j / r_tiles, r = j % r_tiles;
        return ((size_t)blockIdx.x * r_tiles + r) * 32 + 8 * ph + wave;
    };
    u4 ring[kRing];
    unsigned mring[kRing];
    const unsigned short x = vin[threadIdx.x];
    auto issue = [&](int slot, int j) {
        const size_t it = item_of(j < n ? j : n - 1);
        if (j < n) {
            ring[slot] = __builtin_nontemporal_load(w + it * istride + lane);
            if (META == 1) mring[slot] = __builtin_nontemporal_load(meta + it * 16 + (lane & 15));
            if (META == 2) mring[slot] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(w + it * istride + 64) + (lane & 15));
        } else {
            ring[slot] = (u4){0, 0, 0, 0};
            mring[slot] = 0;
        }
    };
#pragma unroll
    for (int s = 0; s < kRing; ++s) issue(s, s);
    unsigned acc = x;
    for (int j0 = 0; j0 < n; j0 += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const u4 v = ring[s];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            if (META) acc += mring[s];
            issue(s, j0 + s + kRing);
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
    if (threadIdx.x < 16) vout[blockIdx.x * 16 + threadIdx.x] = (unsigned short)(x + 1);
}

int main() {
    const int grid = 256, stages = 40, nbuf = 6;
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    const size_t wbytes = (size_t)grid * 8 * 32 * 1088;
    std::vector<u4*> w(nbuf); std::vector<unsigned*> m(nbuf);
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&w[i], wbytes)); CK(hipMemset(w[i], 0x5a, wbytes)); CK(hipMalloc(&m[i], wbytes / 16)); CK(hipMemset(m[i], 1, wbytes / 16)); }
    unsigned short* vec; CK(hipMalloc(&vec, 8192 * 2)); CK(hipMemset(vec, 0, 8192 * 2));
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    auto run = [&](auto kern, const char* name, int r_tiles) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        for (int st = 0; st < stages; ++st)
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kT), 0, s0, w[st % nbuf], m[st % nbuf], r_tiles, vec + (st & 1) * 4096, vec + ((st + 1) & 1) * 4096, sink);
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s0));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double us = best * 1e3 / (5 * stages), mb = (double)grid * r_tiles * 32 * 1024 / 1e6;
        printf("%-34s R=%d (%3d KiB/WG, %5.1f MB): %6.2f us/stage  %5.2f TB/s\n", name, r_tiles, r_tiles * 32, mb, us, mb / us);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    for (int r : {1, 2, 4, 7}) {
        run(k_stream<0, 0>, "contiguous per wave", r);
        run(k_stream<0, 1>, "contiguous + 64 B side stream", r);
        run(k_stream<0, 2>, "contiguous, 1088-byte items", r);
        run(k_stream<1, 0>, "phase order", r);
        run(k_stream<1, 1>, "phase order + 64 B side stream", r);
        run(k_stream<1, 2>, "phase order, 1088-byte items", r);
    }
    return 0;
}
