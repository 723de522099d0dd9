// x_delivery_probe.hip -- how fast can every CU pull the SAME L2-resident activation block into LDS, by how it is moved?
// The slab kernel (w4_slab.hip) gives each wave its own 128-k groups of all rows: at 32 rows every workgroup moves 256 KB of x
// (as much as its weights) L2 -> LDS by LDS-DMA; its timeline shows waves spending 1.5 .. 8 us in the ISSUE of the prologue
// (16 DMAs + 14 weight loads).  Arms, 256 workgroups x 8 waves, every wave moves `kib` KiB of a shared 256 KB block (its own
// 32 KB slice, row-contiguous 1 KiB pieces) into its LDS region, 8 KiB at a time, waits, repeats:
//   dma4   buffer_load_dwordx4 ... lds     (the slab kernel's)
//   dma1   buffer_load_dword ... lds       (four times the instructions)
//   ldst   buffer_load_dwordx4 -> VGPR -> ds_write_b128
//   ld     buffer_load_dwordx4 -> VGPR, no LDS write
// Reported: us per launch inside a hipGraph chain, and bytes per clock and CU at 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/x_delivery_probe tools/ubench/x_delivery_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int ARM>
__global__ __launch_bounds__(512, 2) void k_pull(const unsigned char* x, int pieces, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* reg = smem + wave * 8192;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(x), 0, 256u * 1024u, 0x00020000);
    unsigned acc = 0;
    for (int it = 0; it < pieces; it += 8) {
        const unsigned base = (unsigned)(((wave * 32 + it) & 255) * 1024);    // the wave's slice of the block, wrapping
        if constexpr (ARM == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(reg + i * 1024), 16, (unsigned)lane * 16u, base + i * 1024, 0, 0);
#else
                ;
#endif
            __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));   // vmcnt(0)
        } else if constexpr (ARM == 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i)
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(reg + i * 256), 4, (unsigned)lane * 4u, base + i * 256, 0, 0);
#else
                ;
#endif
            __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
        } else {
            u4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)lane * 16u, base + i * 1024, 0));
            if constexpr (ARM == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<u4*>(reg + i * 1024 + lane * 16) = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].w;
            }
        }
        if (ARM != 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc ^= *reinterpret_cast<unsigned*>(reg + ((it * 64 + lane * 4) & 8188));
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    unsigned char* x; unsigned* sink;
    CK(hipMalloc(&x, 256 * 1024)); CK(hipMalloc(&sink, 64));
    std::vector<unsigned char> h(256 * 1024);
    for (auto& b : h) b = (unsigned char)rand();
    CK(hipMemcpy(x, h.data(), h.size(), hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    const char* names[4] = {"dma4  (buffer_load_dwordx4 lds)", "dma1  (buffer_load_dword lds)  ", "ldst  (load x4 + ds_write_b128)", "ld    (load x4, no LDS write)  "};
    for (int kib : {32, 128}) {
        printf("%d KiB per wave (%d KB per workgroup)\n", kib, kib * 8);
        for (int arm = 0; arm < 4; ++arm) {
            auto launch = [&]() {
                switch (arm) {
                    case 0: hipLaunchKernelGGL(k_pull<0>, dim3(256), dim3(512), 65536, st, x, kib, sink); break;
                    case 1: hipLaunchKernelGGL(k_pull<1>, dim3(256), dim3(512), 65536, st, x, kib, sink); break;
                    case 2: hipLaunchKernelGGL(k_pull<2>, dim3(256), dim3(512), 65536, st, x, kib, sink); break;
                    default: hipLaunchKernelGGL(k_pull<3>, dim3(256), dim3(512), 65536, st, x, kib, sink); break;
                }
            };
            launch(); CK(hipStreamSynchronize(st));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, st));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double us = ms * 1e3 / 100.0, bytes_cu = (double)kib * 1024 * 8;
            printf("  %s : %7.2f us per launch  %6.1f B/clk/CU\n", names[arm], us, bytes_cu / (us * 2400.0));
        }
    }
    return 0;
}
