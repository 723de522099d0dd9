// ldsdma_stream.hip -- what does a per-CU LDS-DMA weight stream deliver on MI355X, by how it is issued?  256 workgroups (one per
// CU), NL loader waves each, every workgroup streams `kib` KiB of its own contiguous slice of a buffer (HBM-cold: the chain
// rotates through `nbuf` buffers > the 256 MB Infinity Cache) into an LDS ring with global_load_lds_dwordx4; no consumers --
// a slot is reused as soon as its DMAs are known to have landed (counted s_waitcnt vmcnt).  Reported: us per launch inside a
// hipGraph chain of dependent launches, TB/s.  Reference arm: the same bytes with plain global_load_dwordx4 into registers
// (8 waves x 8 KiB in flight: the integer-plane kernel's ring).
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_stream ldsdma_stream.hip ; run: ./ldsdma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT>
__device__ __forceinline__ void dma4k(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ const unsigned char* uptr(const unsigned char* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (const unsigned char*)(((uint64_t)hi << 32) | lo);
}

// FLY = 4 KiB pieces a loader keeps in flight (<= 15: 60 DMAs); ring = FLY pieces per loader
// the engine's order: 8 KiB chunks, seven row tiles of 32 KiB each walked chunk-major (tile r, chunk gi) -> r * 32 KiB + gi * 8 KiB
template <int FLY>
__global__ __launch_bounds__(256, 1) void k_dma_tiles(const unsigned char* buf, int kib_per_wg, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 1) return;
    const int pieces = kib_per_wg / 4;
    const unsigned char* base = buf + (size_t)blockIdx.x * kib_per_wg * 1024;
    const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t voff = lane * 16u;
    int issued = 0;
    for (int p = 0; p < pieces; ++p) {
        const int c = p >> 1, gi = c / 7, r = c % 7;
        if (issued >= FLY) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (FLY - 1)) : "memory");
        dma4k<1>(uptr(base + (size_t)r * 32768 + (size_t)gi * 8192 + (size_t)(p & 1) * 4096), voff,
                 __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)(issued % FLY) * 4096u));
        ++issued;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0 && blockIdx.x == 0x7fffffff) sink[0] = *(volatile uint32_t*)smem;
}

template <int FLY, int NT>
__global__ __launch_bounds__(256, 1) void k_dma(const unsigned char* buf, int kib_per_wg, int nl, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nl) return;
    const int pieces = kib_per_wg / 4;                    // 4 KiB pieces of this workgroup
    const unsigned char* base = buf + (size_t)blockIdx.x * kib_per_wg * 1024;
    const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)wave * FLY * 4096u;
    const uint32_t voff = lane * 16u;
    int issued = 0;
    for (int p = wave; p < pieces; p += nl) {
        if (issued >= FLY) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (FLY - 1)) : "memory");   // the oldest piece has landed: its slot is free
        dma4k<NT>(uptr(base + (size_t)p * 4096), voff, __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)(issued % FLY) * 4096u));
        ++issued;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0 && blockIdx.x == 0x7fffffff) sink[0] = *(volatile uint32_t*)smem;
}

// the same loader (one wave, 32 KiB in flight, nt) with EIGHT other waves busy beside it until it is done:
//   MODE 1: polling one LDS word (ds_read_b32 + s_sleep 1)      MODE 2: ds_read_b128 sweeps over the ring (no polling)
//   MODE 3: back-to-back v_mfma_i32_16x16x64_i8                  MODE 4: a VALU loop (v_fma)        MODE 5: 1 + 2 + 3 interleaved
//   MODE 6: s_sleep only (resident, idle)
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE, int BIGLDS>
__global__ __launch_bounds__(576, 1) void k_dma_busy(const unsigned char* buf, int kib_per_wg, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FLY = 8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(smem + FLY * 4096);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (wave == 8) {
        const int pieces = kib_per_wg / 4;
        const unsigned char* base = buf + (size_t)blockIdx.x * kib_per_wg * 1024;
        const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const uint32_t voff = lane * 16u;
        int issued = 0;
        for (int p = 0; p < pieces; ++p) {
            if (issued >= FLY) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (FLY - 1)) : "memory");
            dma4k<1>(uptr(base + (size_t)p * 4096), voff, __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)(issued % FLY) * 4096u));
            ++issued;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *flag = 1;
        return;
    }
    v4i acc = {0, 0, 0, 0}, a = {lane, 1, 2, 3}, b = {4, 5, lane, 7};
    float f = (float)lane;
    uint32_t x = 0;
    for (int it = 0; it < (1 << 20); ++it) {
        if (MODE == 1 || MODE == 5) { if (*flag) break; __builtin_amdgcn_s_sleep(1); }
        else if ((it & 15) == 0 && *flag) break;
        if (MODE == 2 || MODE == 5) {
            const uint4 v = *reinterpret_cast<const uint4*>(smem + ((it & 7) * 4096 + (wave & 3) * 1024 + lane * 16));
            x ^= v.x ^ v.w;
        }
        if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
        }
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 16; ++j) f = __builtin_fmaf(f, 1.0001f, 0.5f);
        }
        if (MODE == 6) __builtin_amdgcn_s_sleep(8);
    }
    if (sink && (acc[0] + (int)x + (int)f) == 0x12345678) sink[lane] = 1;
}

// reference: 8 waves, each 8 x 1 KiB register ring over its share, nt loads
__global__ __launch_bounds__(512, 1) void k_reg(const unsigned char* buf, int kib_per_wg, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int items = kib_per_wg;                         // 1 KiB items
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4* base = reinterpret_cast<const u4*>(buf + (size_t)blockIdx.x * kib_per_wg * 1024) + lane;
    u4 r[8];
    uint32_t acc = 0;
    int it = wave;
#pragma unroll
    for (int j = 0; j < 8; ++j) { r[j] = __builtin_nontemporal_load(base + (size_t)(it < items ? it : wave) * 64); it += 8; }
    for (int c = wave; c < items; c += 64) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
            r[j] = __builtin_nontemporal_load(base + (size_t)(it < items ? it : wave) * 64);
            it += 8;
        }
    }
    if (sink && acc == 0x12345678u) sink[lane] = acc;
}

// wave-private DMA rings: no loader wave, no cross-wave bookkeeping -- each of the 8 waves moves ITS items (item i of the workgroup
// belongs to wave i % 8) through its own ring of D x 1 KiB LDS slots with one global_load_lds_dwordx4 per item, waits with a counted
// vmcnt, reads the item back with ds_read_b128 and refills the slot.  WORK = 1: two i8 MFMAs + a few VALU per item (the integer-plane
// kernel's per-item load), 0: an xor.
template <int D, int WORK>
__global__ __launch_bounds__(512, 1) void k_wave_dma(const unsigned char* buf, int kib_per_wg, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int items = kib_per_wg, n = (items - wave + 7) / 8;
    const unsigned char* base = buf + (size_t)blockIdx.x * kib_per_wg * 1024;
    const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)wave * D * 1024u;
    const uint32_t voff = lane * 16u;
    auto dma1k = [&](int item, int slot) {
        const unsigned char* src = uptr(base + (size_t)(wave + 8 * item) * 1024);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)slot * 1024u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
#pragma unroll
    for (int j = 0; j < D; ++j) if (j < n) dma1k(j, j);
    v4i acc = {0, 0, 0, 0};
    uint32_t x = 0;
    float f = 0.f;
    typedef __attribute__((address_space(3))) const v4i* lds_v4i;
    for (int j = 0; j < n; ++j) {
        if (j + D <= n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const v4i w = *reinterpret_cast<lds_v4i>((uintptr_t)(ring0 + (uint32_t)(j % D) * 1024u + voff));
        if (WORK) {
            v4i b0, b1;
            b0[0] = w[0] & 0x0f0f0f0f; b0[1] = (w[0] >> 4) & 0x0f0f0f0f; b0[2] = w[1] & 0x0f0f0f0f; b0[3] = (w[1] >> 4) & 0x0f0f0f0f;
            b1[0] = w[2] & 0x0f0f0f0f; b1[1] = (w[2] >> 4) & 0x0f0f0f0f; b1[2] = w[3] & 0x0f0f0f0f; b1[3] = (w[3] >> 4) & 0x0f0f0f0f;
            v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0, b1, (v4i){0, 0, 0, 0}, 0, 0, 0);
            d = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, b0, d, 0, 0, 0);
            f = __builtin_fmaf((float)((d[1] << 8) + d[2]), 1.5f, __builtin_fmaf((float)d[0], 0.25f, f));
            acc[0] ^= d[3];
        } else {
            x ^= (uint32_t)(w[0] ^ w[1] ^ w[2] ^ w[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot has been read before it is refilled
        if (j + D < n) dma1k(j + D, j % D);
    }
    if (sink && (acc[0] + (int)x + (int)f) == 0x12345678) sink[lane] = 1;
}

// the register ring with the same per-item work (WORK = 1 above), RD items per wave in flight
template <int RD>
__global__ __launch_bounds__(512, 1) void k_reg_work(const unsigned char* buf, int kib_per_wg, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int items = kib_per_wg;
    const v4i* base = reinterpret_cast<const v4i*>(buf + (size_t)blockIdx.x * kib_per_wg * 1024) + lane;
    v4i r[RD], acc = {0, 0, 0, 0};
    float f = 0.f;
    int it = wave;
#pragma unroll
    for (int j = 0; j < RD; ++j) { r[j] = __builtin_nontemporal_load(base + (size_t)(it < items ? it : wave) * 64); it += 8; }
    for (int c = wave; c < items; c += 8 * RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const v4i w = r[j];
            v4i b0, b1;
            b0[0] = w[0] & 0x0f0f0f0f; b0[1] = (w[0] >> 4) & 0x0f0f0f0f; b0[2] = w[1] & 0x0f0f0f0f; b0[3] = (w[1] >> 4) & 0x0f0f0f0f;
            b1[0] = w[2] & 0x0f0f0f0f; b1[1] = (w[2] >> 4) & 0x0f0f0f0f; b1[2] = w[3] & 0x0f0f0f0f; b1[3] = (w[3] >> 4) & 0x0f0f0f0f;
            v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0, b1, (v4i){0, 0, 0, 0}, 0, 0, 0);
            d = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, b0, d, 0, 0, 0);
            f = __builtin_fmaf((float)((d[1] << 8) + d[2]), 1.5f, __builtin_fmaf((float)d[0], 0.25f, f));
            acc[0] ^= d[3];
            r[j] = __builtin_nontemporal_load(base + (size_t)(it < items ? it : wave) * 64);
            it += 8;
        }
    }
    if (sink && (acc[0] + (int)f) == 0x12345678) sink[lane] = 1;
}

int main() {
    const int nbuf = 6, kibs[] = {32, 224};
    const size_t bytes = 256u * 224u * 1024u;
    std::vector<unsigned char*> bufs(nbuf);
    // RANDOM bytes (packed int4 weights look like this): with a constant fill the same streams ran 10-40 % faster
    std::vector<uint32_t> rnd(bytes / 4);
    uint64_t sd = 88172645463325252ull;
    for (auto& v : rnd) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = (uint32_t)(sd >> 16); }
    const bool constant_fill = getenv("ZL_CONST_FILL") != nullptr;
    for (auto& b : bufs) {
        CK(hipMalloc(&b, bytes));
        if (constant_fill) CK(hipMemset(b, 1, bytes));
        else { CK(hipMemcpy(b, rnd.data(), bytes, hipMemcpyHostToDevice)); rnd[0] += 1; }
    }
    printf("buffers filled with %s\n", constant_fill ? "a constant" : "random bytes");
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
#define ARM(FLY, NT)                                                                                                        \
    for (int nl = 1; nl <= 4; nl *= 2) {                                                                                    \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<FLY, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        if ((size_t)nl * FLY * 4096 > 160 * 1024) continue;                                                                 \
        const float us = time_chain([&](unsigned char* b) { hipLaunchKernelGGL((k_dma<FLY, NT>), dim3(256), dim3(256), nl * FLY * 4096, st, b, kib, nl, sink); }); \
        printf("  LDS-DMA  %d loader wave(s) x %2d KiB in flight%s: %6.2f us  %5.2f TB/s\n", nl, FLY * 4, NT ? " nt" : "   ", us, 256.0 * kib * 1024 / us / 1e6); \
    }
    for (int kib : kibs) {
        printf("%d KiB per workgroup (%.1f MB per launch)\n", kib, 256.0 * kib * 1024 / 1e6);
        const float us = time_chain([&](unsigned char* b) { hipLaunchKernelGGL(k_reg, dim3(256), dim3(512), 0, st, b, kib, sink); });
        printf("  register ring, 8 waves x 8 KiB, nt        : %6.2f us  %5.2f TB/s\n", us, 256.0 * kib * 1024 / us / 1e6);
        ARM(4, 1) ARM(8, 1) ARM(15, 1) ARM(15, 0)
        if (kib == 224) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma_tiles<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const float us3 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL((k_dma_tiles<14>), dim3(256), dim3(256), 14 * 4096, st, b, kib, sink); });
            printf("  LDS-DMA  1 loader x 56 KiB nt, the engine's tile order (7 tiles x 32 KiB, chunk-major): %6.2f us  %5.2f TB/s\n", us3, 256.0 * kib * 1024 / us3 / 1e6);
        }
#define BUSY(MODE, BIG, LABEL)                                                                                             \
        {                                                                                                                  \
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma_busy<MODE, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            const float us2 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL((k_dma_busy<MODE, BIG>), dim3(256), dim3(576), BIG ? 160 * 1024 : 8 * 4096 + 64, st, b, kib, sink); }); \
            printf("  LDS-DMA  1 loader x 32 KiB nt + 8 waves %-34s: %6.2f us  %5.2f TB/s\n", LABEL, us2, 256.0 * kib * 1024 / us2 / 1e6); \
        }
        BUSY(6, 0, "asleep") BUSY(6, 1, "asleep, 160 KiB LDS") BUSY(1, 0, "polling an LDS word") BUSY(2, 0, "reading the ring (ds_read_b128)")
        BUSY(3, 0, "on the i8 matrix cores") BUSY(4, 0, "in a VALU loop") BUSY(5, 0, "poll + read + MFMA") BUSY(5, 1, "poll + read + MFMA, 160 KiB LDS")
        {
#define RREG(RD)                                                                                                            \
            {                                                                                                               \
                const float usw = time_chain([&](unsigned char* b) { hipLaunchKernelGGL(k_reg_work<RD>, dim3(256), dim3(512), 0, st, b, kib, sink); }); \
                printf("  register ring, 8 waves x %d KiB, nt, 2 MFMA + 5 VALU per item         : %6.2f us  %5.2f TB/s\n", RD, usw, 256.0 * kib * 1024 / usw / 1e6); \
            }
            RREG(2) RREG(3) RREG(4) RREG(6) RREG(8)
        }
#define WDMA(DD, WORK, LABEL)                                                                                              \
        {                                                                                                                  \
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_dma<DD, WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            const float us4 = time_chain([&](unsigned char* b) { hipLaunchKernelGGL((k_wave_dma<DD, WORK>), dim3(256), dim3(512), 8 * DD * 1024, st, b, kib, sink); }); \
            printf("  wave-private DMA rings, 8 waves x %2d KiB, nt, %-28s: %6.2f us  %5.2f TB/s\n", DD, LABEL, us4, 256.0 * kib * 1024 / us4 / 1e6); \
        }
        WDMA(2, 1, "2 MFMA + 5 VALU per item") WDMA(3, 1, "2 MFMA + 5 VALU per item") WDMA(4, 0, "xor") WDMA(8, 0, "xor") WDMA(16, 0, "xor") WDMA(4, 1, "2 MFMA + 5 VALU per item") WDMA(8, 1, "2 MFMA + 5 VALU per item") WDMA(16, 1, "2 MFMA + 5 VALU per item")
    }
    return 0;
}
