// How fast does the S-phase pattern of the fused mean-shift kernel run in isolation?  One 32x32 tile, K = 128: 64 dependent
// v_mfma_f32_32x32x2_f32 fed from LDS (two ds_read_b128 per 4 MFMAs).  V1: reads right before their MFMAs (what the
// compiler emits); V2: all 32 fragment reads first, then 64 MFMAs from registers; V3: V1 with two independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 132;

template <int VAR>
__global__ __launch_bounds__(256) void ks(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float sx[64 * LD], sq[64 * LD];
    for (int i = threadIdx.x; i < 64 * LD; i += 256) { sx[i] = 0.001f * (i % 97); sq[i] = 0.002f * (i % 89); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const float *xa = sx + ((wave >> 1) * 32 + li) * LD + lh * 4;
    const float *qb = sq + ((wave & 1) * 32 + li) * LD + lh * 4;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        if (VAR == 2) {
            float4 a[16], b[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) { a[g] = *reinterpret_cast<const float4 *>(xa + g * 8); b[g] = *reinterpret_cast<const float4 *>(qb + g * 8); }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, b[g].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, b[g].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, b[g].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, b[g].w, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a = *reinterpret_cast<const float4 *>(xa + g * 8);
                const float4 b = *reinterpret_cast<const float4 *>(qb + g * 8);
                if (VAR == 3 && (g & 1)) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// VAR 4: the PV-phase pattern alone (A = accumulator registers of the S tile, B = 4 ds_read_b32 per key row, 4 independent
// accumulators); VAR 5: S phase (two accumulators) + exp transform + PV phase, LDS only (no global traffic, no barriers);
// VAR 6: VAR 5 with the single-accumulator S phase of the production kernel
template <int VAR>
__global__ __launch_bounds__(256) void kp(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float sx[64 * LD], sq[64 * LD];
    for (int i = threadIdx.x; i < 64 * LD; i += 256) { sx[i] = 0.001f * (i % 97); sq[i] = 0.002f * (i % 89); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int kh = wave >> 1;
    const float *xa = sx + (kh * 32 + li) * LD + lh * 4;
    const float *qb = sq + ((wave & 1) * 32 + li) * LD + lh * 4;
    const float *xs = sx + (kh * 32 + 4 * lh) * LD + li;
    f32x16 oacc[4], sacc, sacc2;
    for (int d = 0; d < 4; ++d)
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.01f * r; sacc2[r] = 0.f; }
    float rsum = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 5) {
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; sacc2[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a = *reinterpret_cast<const float4 *>(xa + g * 8);
                const float4 b = *reinterpret_cast<const float4 *>(qb + g * 8);
                if (VAR == 5 && (g & 1)) {
                    sacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, sacc2, 0, 0, 0);
                    sacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, sacc2, 0, 0, 0);
                    sacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, sacc2, 0, 0, 0);
                    sacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, sacc2, 0, 0, 0);
                } else {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, sacc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = (sacc[r] + sacc2[r] - 1.0f) * 2.7f;
                e = fminf(fmaxf(e, -13.0f), 75.0f);
                const float p = __expf(e);
                rsum += p;
                sacc[r] = p;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *row = xs + ((r & 3) + 8 * (r >> 2)) * LD;
#pragma unroll
            for (int d = 0; d < 4; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], row[32 * d], oacc[d], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = rsum;
    for (int d = 0; d < 4; ++d)
        for (int r = 0; r < 16; ++r) s += oacc[d][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
static void runp(const char *name, int wgs_per_cu)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL((kp<VAR>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    hipLaunchKernelGGL((kp<VAR>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double mfmas = (double)iters * (VAR >= 5 ? 128 : 64) * 4 * 256 * wgs_per_cu;
    printf("%-44s %d WG/CU: %7.1f TFLOP/s\n", name, wgs_per_cu, mfmas * 4096.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

template <int VAR>
static void run(const char *name, int wgs_per_cu)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL((ks<VAR>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    hipLaunchKernelGGL((ks<VAR>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double mfmas = (double)iters * 64 * 4 * 256 * wgs_per_cu;
    printf("%-44s %d WG/CU: %7.1f TFLOP/s\n", name, wgs_per_cu, mfmas * 4096.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main()
{
    run<1>("V1 reads before their MFMAs", 1); run<1>("V1 reads before their MFMAs", 2);
    run<2>("V2 all reads first, then 64 MFMAs", 1); run<2>("V2 all reads first, then 64 MFMAs", 2);
    run<3>("V3 V1 with two accumulators", 1); run<3>("V3 V1 with two accumulators", 2);
    runp<4>("V4 PV phase alone", 1); runp<4>("V4 PV phase alone", 2);
    runp<5>("V5 S (2 acc) + exp + PV, LDS only", 1); runp<5>("V5 S (2 acc) + exp + PV, LDS only", 2);
    runp<6>("V6 S (1 acc) + exp + PV, LDS only", 1); runp<6>("V6 S (1 acc) + exp + PV, LDS only", 2);
    return 0;
}
