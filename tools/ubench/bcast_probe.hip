// bcast_probe.hip -- what does the BROADCAST operand of a batch-1 GEMV cost?  Every workgroup of a decode GEMV reads the
// same activation vector (8 KB) and, with the fused RMSNorm, the same norm weight (8 KB); the attn_out projection with the
// split merge in its prologue reads the same ~150 KB of attention partials.  A chain of dependent kernels (hipGraph, 256
// workgroups x 8 waves, ring of 1 KiB non-temporal weight loads, no compute) in which every stage first reads XB bytes that
// the PREVIOUS stage wrote (16 halfs per workgroup), laid out with a configurable chunk stride, optionally gates the ring
// consumption on them (LDS + barrier, as the real kernels do), and stamps entry / x arrival / end per workgroup.
//   xmode 0: no broadcast read (one half per thread, as stream_probe)
//   xmode 1: 8 KB vector, 16 B per thread, contiguous
//   xmode 2: 8 KB vector + 8 KB second vector (norm weight), contiguous
//   xmode 3: 8 KB vector, 128-byte chunks `stride` bytes apart (channel-striped)
//   xmode 4: 8 KB + 8 KB, both striped
//   xmode 5: 16 x 512 B per thread of a 256 KB block (the split-merge prologue)
//   xmode 6: 4 KB vector: one wave's worth per ... (half the threads load) -- pre-normalised activations only
// gate 1: the ring is consumed only after x has been written to LDS and a barrier has passed.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bcast_probe tools/ubench/bcast_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kT = 512, kRing = 7;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct P {
    const u4* w;
    int r_tiles;
    const unsigned char* vin;    // the vector the previous stage wrote
    unsigned char* vout;
    const unsigned char* nw;     // second broadcast vector (never written)
    const float* part;           // 256 KB block
    int stride;                  // bytes between 128-byte chunks (xmode 3 / 4)
    unsigned* sink;
    unsigned long long* stamps;  // [stage][wg][4] or null
    int stage;
};

__device__ __forceinline__ size_t striped(int t, int stride) { return (size_t)(t >> 3) * stride + (t & 7) * 16; }

template <int XMODE, int GATE, int BAR = 0>
__global__ __launch_bounds__(kT, 2) void k_stage(const P p) {
    __shared__ u4 xs[2 * kT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    if (p.stamps && threadIdx.x == 0) t0 = wall_clock64();
    const int n = p.r_tiles * 4;
    u4 xv = (u4){0, 0, 0, 0}, nv = (u4){0, 0, 0, 0};
    float pv[16][2];
    if (XMODE == 0) xv.x = reinterpret_cast<const unsigned short*>(p.vin)[threadIdx.x];
    if (XMODE == 1 || XMODE == 2) xv = *reinterpret_cast<const u4*>(p.vin + threadIdx.x * 16);
    if (XMODE == 2) nv = *reinterpret_cast<const u4*>(p.nw + threadIdx.x * 16);
    if (XMODE == 3 || XMODE == 4) xv = *reinterpret_cast<const u4*>(p.vin + striped(threadIdx.x, p.stride));
    if (XMODE == 4) nv = *reinterpret_cast<const u4*>(p.nw + striped(threadIdx.x, p.stride));
    if (XMODE == 6 && threadIdx.x < 256) xv = *reinterpret_cast<const u4*>(p.vin + threadIdx.x * 16);
    if (XMODE == 5) {
        xv.x = reinterpret_cast<const unsigned short*>(p.vin)[threadIdx.x];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float2 f = *reinterpret_cast<const float2*>(p.part + ((size_t)(threadIdx.x >> 4) * 16 + u) * 130 + (threadIdx.x & 15) * 8);
            pv[u][0] = f.x; pv[u][1] = f.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (BAR == 1) __builtin_amdgcn_s_barrier();      // every wave's x loads enter the CU's memory pipeline before any weight load
    if (BAR == 2) __builtin_amdgcn_s_sleep(4);
    if (BAR == 4) __builtin_amdgcn_s_sleep(8);
    if (BAR == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" : "+v"(xv), "+v"(nv)); }
    __builtin_amdgcn_sched_barrier(0);
    auto item_of = [&](int j) -> size_t { return (size_t)(blockIdx.x * 8 + wave) * n + j; };
    u4 ring[kRing];
    auto issue = [&](int slot, int j) {
        if (j < n) ring[slot] = __builtin_nontemporal_load(p.w + item_of(j) * 64 + lane);
        else ring[slot] = (u4){0, 0, 0, 0};
    };
#pragma unroll
    for (int s = 0; s < kRing; ++s) issue(s, s);
    __builtin_amdgcn_sched_barrier(0);
    unsigned acc = 0;
    if (GATE) {
        if (XMODE == 5) {
            float a = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) a += pv[u][0] + pv[u][1];
            xv.y = __builtin_bit_cast(unsigned, a);
        }
        xs[threadIdx.x] = xv;
        xs[kT + threadIdx.x] = nv;
        __syncthreads();
        const u4 o = xs[(threadIdx.x * 7 + 3) & (kT - 1)], o2 = xs[kT + ((threadIdx.x * 5 + 1) & (kT - 1))];
        acc = o.x ^ o.y ^ o.z ^ o.w ^ o2.x;
        if (p.stamps && threadIdx.x == 0) t1 = wall_clock64();
    } else {
        acc = xv.x;   // consumed at the end only
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int j0 = 0; j0 < n; j0 += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const u4 v = ring[s];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            issue(s, j0 + s + kRing);
        }
    }
    if (!GATE) {
        acc ^= xv.y ^ xv.z ^ xv.w ^ nv.x ^ nv.y ^ nv.z ^ nv.w;
        if (XMODE == 5) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc ^= __builtin_bit_cast(unsigned, pv[u][0]) ^ __builtin_bit_cast(unsigned, pv[u][1]);
        }
    }
    if (acc == 0x12345678u) p.sink[threadIdx.x] = acc;
    // this workgroup's 16 halfs of the next stage's vector (32 bytes), in the layout the next stage reads
    if (threadIdx.x < 2) {
        const int t = blockIdx.x * 2 + threadIdx.x;   // 16-byte unit
        const size_t off = (XMODE == 3 || XMODE == 4) ? striped(t, p.stride) : (size_t)t * 16;
        *reinterpret_cast<u4*>(p.vout + off) = (u4){acc, 1, 2, 3};
    }
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long* s = p.stamps + ((size_t)p.stage * 256 + blockIdx.x) * 4;
        s[0] = t0; s[1] = t1; s[2] = wall_clock64();
    }
}

int main(int argc, char** argv) {
    const int grid = 256, stages = 40, nbuf = 6;
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    const size_t wbytes = (size_t)grid * 8 * 32 * 1024;
    std::vector<u4*> w(nbuf);
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&w[i], wbytes)); CK(hipMemset(w[i], 0x5a, wbytes)); }
    const size_t vbytes = 64 * 16384 + 8192;   // room for 64 chunks at up to 16 KB stride
    unsigned char* vec[2]; for (auto& v : vec) { CK(hipMalloc(&v, vbytes)); CK(hipMemset(v, 0, vbytes)); }
    unsigned char* nw; CK(hipMalloc(&nw, vbytes)); CK(hipMemset(nw, 1, vbytes));
    float* part; CK(hipMalloc(&part, 32 * 16 * 130 * 4 + 4096)); CK(hipMemset(part, 0, 32 * 16 * 130 * 4 + 4096));
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)stages * 256 * 4 * 8));
    auto run = [&](auto kern, const char* name, int r_tiles, int stride) {
        for (int pass = 0; pass < 2; ++pass) {       // pass 0: timing, pass 1: stamps
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
            for (int st = 0; st < stages; ++st) {
                P p{w[st % nbuf], r_tiles, vec[st & 1], vec[(st + 1) & 1], nw, part, stride, sink, pass ? stamps : nullptr, st};
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kT), 0, s0, p);
            }
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
            if (pass == 0) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0, s0));
                    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s0));
                    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                printf("%-44s R=%d stride %5d: %6.2f us/stage", name, r_tiles, stride, best * 1e3 / (5 * stages));
            } else {
                CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
                std::vector<unsigned long long> h((size_t)stages * 256 * 4);
                CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
                std::vector<double> xa, dur;
                for (int st = 4; st < stages; ++st) {
                    unsigned long long first = ~0ull;
                    for (int b = 0; b < 256; ++b) first = std::min(first, h[((size_t)st * 256 + b) * 4]);
                    for (int b = 0; b < 256; ++b) {
                        const unsigned long long* s = &h[((size_t)st * 256 + b) * 4];
                        if (s[1]) xa.push_back((s[1] - first) * 0.01);
                        dur.push_back((s[2] - first) * 0.01);
                    }
                }
                std::sort(xa.begin(), xa.end()); std::sort(dur.begin(), dur.end());
                if (!xa.empty()) printf("   x ready p50 %.2f p90 %.2f", xa[xa.size() / 2], xa[xa.size() * 9 / 10]);
                printf("   wg end p50 %.2f max %.2f us\n", dur[dur.size() / 2], dur.back());
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    };
    (void)argc; (void)argv;
    for (int r : {0, 1, 2, 7}) {
        run(k_stage<0, 0>, "no broadcast", r, 0);
        run(k_stage<0, 1>, "1 KB, gated", r, 0);
        run(k_stage<1, 0>, "8 KB contiguous, ungated", r, 0);
        run(k_stage<1, 1>, "8 KB contiguous, gated", r, 0);
        run(k_stage<2, 1>, "8+8 KB contiguous, gated", r, 0);
        run(k_stage<1, 1, 1>, "8 KB contiguous, gated, barrier before ring", r, 0);
        run(k_stage<2, 1, 1>, "8+8 KB contiguous, gated, barrier before ring", r, 0);
        run(k_stage<1, 1, 2>, "8 KB contiguous, gated, sleep 4 before ring", r, 0);
        run(k_stage<1, 1, 4>, "8 KB contiguous, gated, sleep 8 before ring", r, 0);
        run(k_stage<1, 1, 3>, "8 KB contiguous, gated, x waited before ring", r, 0);
        run(k_stage<2, 1, 3>, "8+8 KB contiguous, gated, x waited before ring", r, 0);
        run(k_stage<5, 0>, "256 KB partials (merge), ungated", r, 0);
        run(k_stage<5, 1>, "256 KB partials (merge), gated", r, 0);
    }
    return 0;
}
