// Pure-register fp32 MFMA throughput on gfx950: how fast can v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 issue with
// NACC independent accumulators per wave and WPS waves per SIMD?  (tools/micro: measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char *name, F launch, double flop_per_mfma, int nacc, int wgs_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 4000;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    launch(out, 10, 256 * wgs_per_cu);
    hipDeviceSynchronize();
    hipEventRecord(s);
    launch(out, iters, 256 * wgs_per_cu);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double mfmas = (double)iters * 8 * nacc * 4 /*waves*/ * 256 * wgs_per_cu;
    printf("%-34s %d acc, %d waves/SIMD: %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", name, nacc, wgs_per_cu,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfmas / (1024.0)));
    hipFree(out);
}

int main()
{
#define R32(N, W) run("v_mfma_f32_32x32x2_f32", [](float *o, int it, int g) { hipLaunchKernelGGL((k32<N>), dim3(g), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 4096.0, N, W)
#define R16(N, W) run("v_mfma_f32_16x16x4_f32", [](float *o, int it, int g) { hipLaunchKernelGGL((k16<N>), dim3(g), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 2048.0, N, W)
    R32(1, 1); R32(2, 1); R32(4, 1); R32(1, 2); R32(4, 2); R32(4, 4);
    R16(1, 1); R16(2, 1); R16(4, 1); R16(8, 1); R16(4, 2); R16(8, 2);
    return 0;
}
