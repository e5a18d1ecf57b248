// What does moving a k-tile into LDS cost a wave that is otherwise issuing fp32 MFMAs?  The tiled GEMM's ratio: per 32
// MFMAs (32x32x2) a wave issues 4 x 16-byte loads, 4 x ds_write_b128 and 12 x ds_read_b128.  Bare loops, wave-private LDS
// regions (no barriers), loads from an L2-resident window, two waves per SIMD.
//   0: MFMAs only (operands in registers)      1: + 12 ds_read_b128 feeding the MFMAs
//   2: 1 + 4 buffer_load_dwordx4 -> VGPR        3: 2 + 4 ds_write_b128 (the kernel's staging path)
//   4: 1 + 4 buffer_load_dwordx4 ... lds        (LDS-direct: no VGPR round trip, no ds_write)
// (tools/micro: measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const float *src, int window_bytes, float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 4096];   // per wave: two stages of 4 KB + the region the fragments are read from
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float *mine = lds + wave * 4096;
    for (int i = lane; i < 4096; i += 64) mine[i] = 0.001f * (i & 15);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, window_bytes, 0x00020000);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 op[12];
    for (int i = 0; i < 12; ++i) op[i] = make_float4(0.1f * i, 0.2f, -0.3f, 0.05f * i);
    u32x4 ld[4];
    for (int i = 0; i < 4; ++i) ld[i] = u32x4{0u, 0u, 0u, 0u};
    int soff = ((blockIdx.x * 4 + wave) * 4096) % window_bytes;
    for (int it = 0; it < iters; ++it) {
        float *cur = mine + (it & 1) * 1024, *nxt = mine + ((it + 1) & 1) * 1024;
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ld[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, soff + j * 1024, 0);
        }
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(nxt + j * 256), 16, lane * 16, soff + j * 1024, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < 12; ++j) op[j] = *reinterpret_cast<const float4 *>(mine + (it & 1) * 1024 + (j * 64 + lane) * 4);
        }
        if (MODE == 5 || MODE == 6) {   // dependent chains: all 32 MFMAs into ONE accumulator / alternating between two
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 a = op[j % 4], b = op[4 + j];
                constexpr int W = MODE == 5 ? 0 : 1;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0], 0, 0, 0);
                acc[W] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[W], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[0], 0, 0, 0);
                acc[W] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[W], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 a = op[j % 4];
            const float4 b = op[4 + j];
            if (MODE == 2 && j >= 4)   // the loaded values are used, without VALU work, late enough for their latency
                a = make_float4(__uint_as_float(ld[j - 4].x), __uint_as_float(ld[j - 4].y), __uint_as_float(ld[j - 4].z), __uint_as_float(ld[j - 4].w));
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j & 3], 0, 0, 0);
            acc[(j + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[(j + 1) & 3], 0, 0, 0);
            acc[(j + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[(j + 2) & 3], 0, 0, 0);
            acc[(j + 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[(j + 3) & 3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4 *>(nxt + j * 256 + lane * 4) = ld[j];
        }
        if (MODE == 4) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the LDS-direct loads have landed
        soff += 4096 * 7;
        if (soff >= window_bytes) soff -= window_bytes;
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum + mine[lane];
}

template <int MODE>
static void run(const char *name, const float *src, int window)
{
    float *out;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    const int iters = 20000, launches = 20;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(256), 0, 0, src, window, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(256), 0, 0, src, window, out, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double flop = (double)launches * iters * 32 * 4096.0 * 4 * 512;
    printf("%-64s %5.2f s: %7.1f TFLOP/s\n", name, ms * 1e-3, flop / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    const int window = 1 << 22;   // 4 MB: L2-resident
    float *src;
    hipMalloc(&src, window);
    hipMemset(src, 0, window);
    run<0>("0: 32 MFMAs per iteration, operands in registers", src, window);
    run<1>("1: + 12 ds_read_b128", src, window);
    run<2>("2: + 12 ds_read_b128 + 4 buffer_load_dwordx4 -> VGPR", src, window);
    run<3>("3: + 12 ds_read_b128 + 4 loads -> VGPR + 4 ds_write_b128", src, window);
    run<4>("4: + 12 ds_read_b128 + 4 buffer_load_dwordx4 ... lds", src, window);
    run<5>("5: as 0, all MFMAs into ONE accumulator (dependent chain)", src, window);
    run<6>("6: as 0, two accumulators alternating", src, window);
    return 0;
}
