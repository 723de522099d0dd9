// xmix_probe.hip -- does a broadcast (L2-resident) activation stream cost a CU what an HBM stream costs when both share a wave's
// in-order load queue?  256 workgroups (one per CU) stream 224 KiB of private, HBM-cold weights each (8 waves x register ring of
// 8 x 1 KiB, nt) and read `xkib` KiB of ONE buffer that all workgroups share (the activations of a 5..32-row decode batch):
//   mode 0: weights only
//   mode 1: every wave interleaves its share of the shared buffer with its weight items (same queue: the phase kernel's scheme)
//   mode 2: a ninth wave reads the shared buffer on its own (its own queue), the eight stream weights
//   mode 3: the shared buffer only (eight waves)
// us per launch in a hipGraph chain rotating over 6 weight buffers.  build: hipcc --offload-arch=gfx950 -O3 -o xmix_probe xmix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(576, 1) void k_mix(const unsigned char* wbuf, const unsigned char* xbuf, int wkib, int xkib, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (wave == 8) {
        if (MODE == 2) {
            const v4i* xb = reinterpret_cast<const v4i*>(xbuf) + lane;
            v4i r[8];
            int it = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { r[j] = xb[(size_t)(it < xkib ? it : 0) * 64]; ++it; }
            for (int c = 0; c < xkib; c += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc ^= r[j][0] ^ r[j][3]; r[j] = xb[(size_t)(it < xkib ? it : 0) * 64]; ++it; }
            }
        }
    } else {
        const v4i* wb = reinterpret_cast<const v4i*>(wbuf + (size_t)blockIdx.x * wkib * 1024) + lane;
        const v4i* xb = reinterpret_cast<const v4i*>(xbuf) + lane;
        if (MODE == 3) {
            v4i r[8];
            int it = wave;
#pragma unroll
            for (int j = 0; j < 8; ++j) { r[j] = xb[(size_t)(it < xkib ? it : wave) * 64]; it += 8; }
            for (int c = wave; c < xkib; c += 64) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc ^= r[j][0] ^ r[j][3]; r[j] = xb[(size_t)(it < xkib ? it : wave) * 64]; it += 8; }
            }
        } else {
            v4i r[8], xr[8];
            int it = wave, xit = wave;
            // x items per weight item for this wave (xkib : wkib), issued right behind the weight item
            const int xper = (xkib * 8 + wkib - 1) / wkib;     // in eighths
#pragma unroll
            for (int j = 0; j < 8; ++j) { r[j] = __builtin_nontemporal_load(wb + (size_t)(it < wkib ? it : wave) * 64); it += 8; xr[j] = (v4i){0, 0, 0, 0}; }
            int xcred = 0;
            for (int c = wave; c < wkib; c += 64) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc ^= r[j][0] ^ r[j][3] ^ xr[j][1];
                    r[j] = __builtin_nontemporal_load(wb + (size_t)(it < wkib ? it : wave) * 64);
                    it += 8;
                    if (MODE == 1) {
                        xcred += xper;
                        if (xcred >= 8) { xcred -= 8; xr[j] = xb[(size_t)(xit < xkib ? xit : wave) * 64]; xit += 8; }
                    }
                }
            }
        }
    }
    if (sink && acc == 0x12345678u) sink[lane] = acc;
}

int main() {
    const int nbuf = 6, wkib = 224;
    const size_t bytes = 256u * 224u * 1024u;
    std::vector<unsigned char*> bufs(nbuf);
    std::vector<uint32_t> rnd(bytes / 4);
    uint64_t sd = 88172645463325252ull;
    for (auto& v : rnd) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = (uint32_t)(sd >> 16); }
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemcpy(b, rnd.data(), bytes, hipMemcpyHostToDevice)); rnd[0] += 1; }
    unsigned char* xbuf; CK(hipMalloc(&xbuf, 1 << 20)); CK(hipMemcpy(xbuf, rnd.data(), 1 << 20, hipMemcpyHostToDevice));
    uint32_t* sink; CK(hipMalloc(&sink, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto time_chain = [&](auto launch) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 24; ++i) launch(bufs[i % nbuf]);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return ms * 1e3 / (5 * 24);
    };
    for (int xkib : {64, 128, 256, 512}) {
        const float t0 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL(k_mix<0>, dim3(256), dim3(576), 0, st, b, xbuf, wkib, xkib, sink); });
        const float t1 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL(k_mix<1>, dim3(256), dim3(576), 0, st, b, xbuf, wkib, xkib, sink); });
        const float t2 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL(k_mix<2>, dim3(256), dim3(576), 0, st, b, xbuf, wkib, xkib, sink); });
        const float t3 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL(k_mix<3>, dim3(256), dim3(576), 0, st, b, xbuf, wkib, xkib, sink); });
        printf("224 KiB of weights per workgroup + %3d KiB shared: weights only %6.2f us | interleaved in the same waves %6.2f | a ninth wave %6.2f | shared only %6.2f\n",
               xkib, t0, t1, t2, t3);
    }
    return 0;
}
