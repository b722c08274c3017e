// Sustained v_mfma_f32_16x16x32_bf16 rate of the GPU this runs on: nothing but MFMAs on register operands, for the
// context of the roofline fractions in DESIGN.md (the 2.5 PFLOP/s peak is the 2.4 GHz figure; under sustained MFMA load
// the chip clocks lower).  Usage: mfma_peak [waves per SIMD = 2] [independent accumulators per wave = 8]
// Build: make -C video-classification_amd/csrc tools   (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(float* out, int iters, long long* clocks) {
    extern __shared__ char lds_pad[];  // 96 KB per workgroup: exactly one workgroup per CU, its waves spread over the 4 SIMDs
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x & 3), b[i] = (__bf16)(float)(i & 1);
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}

template <int NACC>
static void run(int wps, int iters) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus;        // one workgroup of 4 * wps waves per CU (the LDS request keeps a second one away)
    const int threads = 256 * wps;
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute((const void*)mfma_loop<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* out;
    long long* clk;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipMalloc(&clk, sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), lds, 0, out, iters / 10, clk);  // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    long long hclk = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), lds, 0, out, iters, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms, hipMemcpy(&hclk, clk, sizeof(hclk), hipMemcpyDeviceToHost);
    }
    const double mfmas = (double)blocks * 4 * wps * (double)iters * NACC;
    const double tflops = mfmas * 2.0 * 16 * 16 * 32 / (best * 1e-3) / 1e12;
    printf("CUs %d, %d wave(s) per SIMD, %d accumulators: %.3f ms, %.0f TFLOP/s; s_memtime ticks per MFMA and SIMD %.2f, "
           "ticks per us %.0f\n", cus, wps, NACC, best, tflops, (double)hclk / ((double)iters * NACC * wps), hclk / (best * 1e3));
    hipFree(out), hipFree(clk);
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int nacc = argc > 2 ? atoi(argv[2]) : 8;
    const int iters = 200000;
    if (nacc == 1) run<1>(wps, iters);
    else if (nacc == 2) run<2>(wps, iters);
    else if (nacc == 4) run<4>(wps, iters);
    else if (nacc == 16) run<16>(wps, iters / 2);
    else run<8>(wps, iters);
    return 0;
}
