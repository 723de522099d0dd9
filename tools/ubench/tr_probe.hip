// probe of ds_read_b64_tr_b16 lane semantics (gfx950): prints, per lane, which LDS elements the 4 results came from.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;   // value = its own index
    __syncthreads();
    const int lane = threadIdx.x;
    const int i = lane & 15, kq = lane >> 4;
    // a [64 rows][64 halfs] row-major image; group kq addresses rows 4kq .. 4kq+3, columns 0..15:
    // lane i supplies the address of row 4kq + i/4, columns 4 (i%4) .. +3
    unsigned short* addr = lds + (4 * kq + i / 4) * 64 + 4 * (i % 4);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (r%2d,c%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
