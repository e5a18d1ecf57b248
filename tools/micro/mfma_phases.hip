// Can the second wave of a SIMD fill the first one's non-MFMA phases?  Every wave loops over {64 fp32 MFMAs (4096 pipe
// cycles), a ~2500-cycle phase without MFMAs}: one wave alone can keep the pipe 62 % busy, two waves with complementary
// phases 100 %, two waves in step 62-70 %.  Phase kinds: 1 s_sleep, 2 VALU work, 3 LDS round trips, 4 as 1 with two
// workgroup barriers (4 waves).  OFFSET: the second workgroup of a CU starts half a period late.
// (tools/micro: measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PHASE, bool OFFSET>
#ifndef WPE
#define WPE 2      // waves per SIMD the kernel is compiled for (hipcc -DWPE=4: the 3- and 4-wave rows)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(float *out, int iters, int second_from)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float *mine = lds + wave * 1024;
    for (int i = lane; i < 1024; i += 64) mine[i] = 0.001f * (i & 15);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = 0.37f + lane * 1e-3f, bv = -0.21f;
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = 0.5f + i;
    if (OFFSET && (int)blockIdx.x >= second_from)
        for (int i = 0; i < 26; ++i) __builtin_amdgcn_s_sleep(2);   // ~3300 cycles: half a period
#ifdef PRIO
    // the FIRST workgroup of a CU (dispatch order) runs at a higher wave priority for its whole life: the two waves of a
    // SIMD then cannot fall into step (equal priorities share the pipe, finish their MFMA phases together and idle together)
    if ((int)blockIdx.x < second_from) __builtin_amdgcn_s_setprio(3);
#endif
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (PHASE == 1 || PHASE == 4) {
            if (PHASE == 4) __syncthreads();
#pragma unroll
            for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(2);   // 20 x 128 cycles
            if (PHASE == 4) __syncthreads();
        } else if (PHASE == 2) {
#pragma unroll
            for (int i = 0; i < 75; ++i)
#pragma unroll
                for (int u = 0; u < 8; ++u) va[u] = fmaf(va[u], 1.0001f, 0.25f);   // 600 independent-ish VALU instructions
        } else if (PHASE == 3) {
            float4 v = make_float4(va[0], va[1], va[2], va[3]);
#pragma unroll
            for (int i = 0; i < 20; ++i) {   // 20 dependent LDS round trips
                *reinterpret_cast<float4 *>(mine + lane * 4 + (i & 3) * 256) = v;
                v = *reinterpret_cast<const float4 *>(mine + ((lane + 1) & 63) * 4 + (i & 3) * 256);
            }
            va[0] = v.x;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum + mine[lane];
}

template <int PHASE, bool OFFSET>
static void run(const char *name, int wgs)
{
    float *out;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    const int iters = 4000, launches = 10;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<PHASE, OFFSET>), dim3(wgs), dim3(256), 0, 0, out, iters, 256);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<PHASE, OFFSET>), dim3(wgs), dim3(256), 0, 0, out, iters, 256);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double flop = (double)launches * iters * 64 * 4096.0 * 4 * wgs;
    printf("%-44s %d waves/SIMD%s  %5.2f s: %7.1f TFLOP/s\n", name, wgs / 256, OFFSET ? ", second half a period late" : "                           ",
           ms * 1e-3, flop / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
#if WPE > 2
    for (int w = 1; w <= WPE; ++w) {
        run<1, false>("+ 2560 cycles of s_sleep", 256 * w);
        run<3, false>("+ 20 dependent LDS round trips", 256 * w);
        run<2, false>("+ 600 VALU instructions", 256 * w);
    }
    return 0;
#endif
    run<0, false>("MFMAs only", 256);
    run<0, false>("MFMAs only", 512);
    run<1, false>("+ 2560 cycles of s_sleep", 256);
    run<1, false>("+ 2560 cycles of s_sleep", 512);
    run<1, true>("+ 2560 cycles of s_sleep", 512);
    run<2, false>("+ 600 VALU instructions", 256);
    run<2, false>("+ 600 VALU instructions", 512);
    run<2, true>("+ 600 VALU instructions", 512);
    run<3, false>("+ 20 dependent LDS round trips", 256);
    run<3, false>("+ 20 dependent LDS round trips", 512);
    run<3, true>("+ 20 dependent LDS round trips", 512);
    run<4, false>("+ sleep between two workgroup barriers", 256);
    run<4, false>("+ sleep between two workgroup barriers", 512);
    run<4, true>("+ sleep between two workgroup barriers", 512);
    return 0;
}
