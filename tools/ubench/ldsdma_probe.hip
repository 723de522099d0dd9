// ldsdma_probe.hip -- which LDS addresses can global_load_lds_dwordx4 reach on gfx950 (160 KiB LDS)?  M0 carries the wave-uniform
// destination; if only its low 16 bits counted, a destination past 64 KiB would alias into the first 64 KiB.
// build: hipcc --offload-arch=gfx950 -O2 -o ldsdma_probe ldsdma_probe.hip ; run: ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// saddr form with an instruction offset: does `offset:` move BOTH the global source and the LDS destination?
__device__ __forceinline__ void dma16_off(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__global__ void k_probe_off(const uint32_t* src, uint32_t* out, uint32_t dst_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = lane; i < 160 * 1024 / 4; i += 64) l32[i] = 0xdeadbeefu;
    __syncthreads();
    dma16_off(src, (uint32_t)lane * 16u, __builtin_amdgcn_readfirstlane(dst_off));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int first = -1, cnt = 0;
    for (int i = 0; i < 160 * 1024 / 4; ++i) {
        if (l32[i] != 0xdeadbeefu) { if (first < 0) first = i; ++cnt; }
    }
    if (lane == 0) { out[0] = (uint32_t)first; out[1] = (uint32_t)cnt; out[2] = first >= 0 ? l32[first] : 0u; out[3] = 0; }
}

__global__ void k_probe(const uint32_t* src, uint32_t* out, uint32_t dst_off, int half) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = lane; i < 160 * 1024 / 4; i += 64) l32[i] = 0xdeadbeefu;
    __syncthreads();
    if (!half || lane < 32) dma16(reinterpret_cast<const unsigned char*>(src) + lane * 16, __builtin_amdgcn_readfirstlane(dst_off));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report: first LDS dword index that is not the fill pattern, count of such dwords, and the data at dst_off
    int first = -1, cnt = 0;
    for (int i = 0; i < 160 * 1024 / 4; ++i) {
        if (l32[i] != 0xdeadbeefu) { if (first < 0) first = i; ++cnt; }
    }
    if (lane == 0) { out[0] = (uint32_t)first; out[1] = (uint32_t)cnt; out[2] = l32[dst_off / 4]; out[3] = l32[dst_off / 4 + 255]; }
}

int main() {
    uint32_t *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 64);
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x10000u + i;
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const uint32_t offs[] = {0, 4096, 60 * 1024, 64 * 1024, 64 * 1024 + 4096, 100 * 1024, 128 * 1024, 159 * 1024};
    for (int half = 0; half < 2; ++half)
        for (uint32_t o : offs) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, o, half);
            uint32_t r[4];
            hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
            printf("dst %6u B (%s): first changed dword at byte %d, %u dwords changed, lds[dst]=%#x lds[dst+1020]=%#x  %s\n", o,
                   half ? "lanes 0..31" : "64 lanes", (int)r[0] * 4, r[1], r[2], r[3],
                   ((int)r[0] * 4 == (int)o && r[1] == (half ? 128u : 256u)) ? "OK" : "UNEXPECTED");
        }
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe_off), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (uint32_t o : {0u, 70u * 1024u}) {
        hipLaunchKernelGGL(k_probe_off, dim3(1), dim3(64), 160 * 1024, 0, src, out, o);
        uint32_t r[4];
        hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
        printf("saddr + offset:1024, M0 = %u: first changed byte %d (LDS moved by %d), %u dwords, first value %#x (source moved by %d bytes)\n", o,
               (int)r[0] * 4, (int)r[0] * 4 - (int)o, r[1], r[2], ((int)r[2] - 0x10000) * 4);
    }
    return 0;
}
