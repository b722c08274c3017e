// micro-experiment: 16-B global loads and LDS-DMA loads from 8-byte-aligned addresses
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __attribute__((aligned(128))) unsigned int zero_page[32];

__global__ void k_plain(const unsigned short* src, u32x4* out, int shift) {
    // lane reads 16 B starting at element (lane*12 + shift): 8-byte aligned only when shift%4==0
    const unsigned short* p = src + threadIdx.x * 12 + shift;
    out[threadIdx.x] = *(const u32x4*)p;
}
__global__ void k_glds(const unsigned short* src, u32x4* out, int shift) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 16 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned short* p = src + threadIdx.x * 12 + shift;
    if ((lane & 7) == 3) p = (const unsigned short*)zero_page;  // masked lanes fetch zeros
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, 0, 0);
    __syncthreads();
    out[threadIdx.x] = *(u32x4*)(lds + threadIdx.x * 16);
}
int main() {
    const int n = 256 * 12 + 64;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i * 7 + 1);
    unsigned short* d; u32x4* o; hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 16);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    unsigned short ho[256 * 8];
    for (int variant = 0; variant < 2; ++variant)
        for (int shift = 0; shift <= 4; shift += 4) {
            hipMemset(o, 0xff, 256 * 16);
            if (variant == 0) hipLaunchKernelGGL(k_plain, dim3(1), dim3(256), 0, 0, d, o, shift);
            else hipLaunchKernelGGL(k_glds, dim3(1), dim3(256), 0, 0, d, o, shift);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
            int bad = 0;
            for (int t = 0; t < 256; ++t)
                for (int j = 0; j < 8; ++j) {
                    unsigned short want = h[t * 12 + shift + j];
                    if (variant == 1 && (t & 7) == 3) want = 0;
                    if (ho[t * 8 + j] != want) ++bad;
                }
            printf("variant %s shift %d: err=%d bad=%d\n", variant ? "glds" : "plain", shift, (int)e, bad);
        }
    return 0;
}
