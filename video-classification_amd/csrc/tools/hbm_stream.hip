// Sustained HBM rate of the GPU this runs on, for the context of the bandwidth rooflines in DESIGN.md (SURVEY 8(d):
// "confirm both peaks on the box").  Three streaming kernels over buffers far larger than the 256 MB Infinity Cache:
// read (16-byte loads, summed), write (16-byte stores), copy (one buffer into another: bytes = read + written).
// Usage: hbm_stream [GiB per buffer = 2]
// Build: make -C video-classification_amd/csrc tools   (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int UNROLL = 4;  // independent 16-byte accesses per thread and trip

__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ a, long n, unsigned* out) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = 0;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(&a[i + u * stride]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    for (; i < n; i += stride) s += a[i][0];
    if (s == 0x12345678u) out[0] = s;  // (keeps the loads; practically never true)
}

__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ a, long n, unsigned val) {
    const long stride = (long)gridDim.x * blockDim.x;
    const u32x4 v = {val, val + 1, val + 2, val + 3};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, &a[i]);
}

__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ a, u32x4* __restrict__ b, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(&a[i + u * stride]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], &b[i + u * stride]);
    }
    for (; i < n; i += stride) b[i] = a[i];
}

template <typename F>
static double best_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch();  // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0), hipEventDestroy(e1);
    return best;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const long n = (long)(gib * 1024.0 * 1024.0 * 1024.0 / 16.0);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    u32x4 *a, *b;
    unsigned* out;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) {
        fprintf(stderr, "allocation of 2 x %.1f GiB failed\n", gib);
        return 1;
    }
    hipMemset(a, 1, n * 16), hipMemset(b, 2, n * 16);
    const double bytes = (double)n * 16.0;
    for (int wgs_per_cu : {4, 8, 16}) {
        const int grid = prop.multiProcessorCount * wgs_per_cu;
        const double tr = best_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
        const double tw = best_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, 7u); });
        const double tc = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
        printf("%s, %d CUs, %.1f GiB buffers, %d workgroups per CU: read %.0f GB/s, write %.0f GB/s, copy %.0f GB/s "
               "(read + written)\n", prop.name, prop.multiProcessorCount, gib, wgs_per_cu, bytes / tr / 1e6, bytes / tw / 1e6,
               2.0 * bytes / tc / 1e6);
    }
    hipFree(a), hipFree(b), hipFree(out);
    return 0;
}
