// Do the two fp32 MFMA shapes sustain the same FLOP/s on RANDOM operands under load (the chip lowers its clock under load,
// and the clock it holds can depend on the MFMA shape: /opt/skills/guides/MI355X_MICROARCH.md, DVFS give-back item 7)?
// Bare register loops, operands re-randomised every iteration (xorshift bits -> float in [1, 2)), ~2 s per shape.
// (tools/micro: measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned &s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return __uint_as_float(0x3f800000u | (s >> 9)) - 1.5f;   // [-0.5, 0.5)
}

// MODE 0: random operands; 1: the same VALU work, constant operands (the random values only feed a checksum); 2: no VALU
template <int SHAPE, int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned seed, unsigned long long *clk)
{
    float junk = 0.f;
    f32x16 a32[NACC];
    f32x4 a16[2 * NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) a32[i][r] = 0.f;
    for (int i = 0; i < 2 * NACC; ++i)
        for (int r = 0; r < 4; ++r) a16[i][r] = 0.f;
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u + 1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) { av[u] = rnd(s); bv[u] = rnd(s); }
            else if (MODE == 1) { junk += rnd(s); junk += rnd(s); av[u] = 0.37f + u; bv[u] = -0.21f * (u + 1); }
            else { av[u] = 0.37f + u; bv[u] = -0.21f * (u + 1); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SHAPE == 32) {
#pragma unroll
                for (int i = 0; i < NACC; ++i) a32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + i) & 3], a32[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2 * NACC; ++i) a16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[(u + i) & 3], a16[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = junk;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) sum += a32[i][r];
    for (int i = 0; i < 2 * NACC; ++i)
        for (int r = 0; r < 4; ++r) sum += a16[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, int MODE>
static void run(const char *name)
{
    constexpr int NACC = 4;
    float *out; unsigned long long *clk;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    const int iters = 20000, launches = 60;   // each launch: 20000 x 4 x 4 (or 8) MFMAs per wave
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<SHAPE, NACC, MODE>), dim3(512), dim3(256), 0, 0, out, iters, w, clk);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<SHAPE, NACC, MODE>), dim3(512), dim3(256), 0, 0, out, iters, 100 + l, clk);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)launches * iters * 4 * NACC * 4096.0 * 4 /*waves*/ * 512;
    printf("%-26s %-34s 2 waves/SIMD, %5.2f s: %7.1f TFLOP/s, in-kernel clock %.2f GHz\n", name,
           MODE == 0 ? "random operands," : (MODE == 1 ? "constant operands + the RNG VALU," : "constant operands, no VALU,"), ms * 1e-3,
           flop / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 0.1);
}

int main()
{
    run<32, 0>("v_mfma_f32_32x32x2_f32");
    run<32, 1>("v_mfma_f32_32x32x2_f32");
    run<32, 2>("v_mfma_f32_32x32x2_f32");
    run<16, 0>("v_mfma_f32_16x16x4_f32");
    run<16, 1>("v_mfma_f32_16x16x4_f32");
    run<16, 2>("v_mfma_f32_16x16x4_f32");
    return 0;
}
