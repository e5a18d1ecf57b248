// Stand-alone variants bench of the fused mean-shift FORWARD kernel (csrc/meanshift_fused.hip, MODE 0, FAST) at
// B = 24, N = 2048, D = 128 with its real global traffic.  Build: hipcc -O3 --offload-arch=gfx950 -o msf_variants msf_variants.hip
//   QREG   query fragments (B operand of the S phase) live in registers for the whole kernel (no s_q reads in the loop)
//   DBUF   two LDS buffers for the X tile, ONE barrier per step (s_q is dropped: the epilogue re-reads Z from global)
//   SORD   0: K^T stores at the start of the next step, ahead of the tile loads (production); 1: after the tile loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 128, QB = 64, KB = 64, LDSW = D + 4;
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
typedef float f32x4b __attribute__((ext_vector_type(4)));
struct Tile64 {
    float4 v[8];
    // buffer form: SGPR base, one constant VGPR offset per thread, scalar offset per (step, p)
    __device__ __forceinline__ void loadb(__amdgpu_buffer_rsrc_t rs, int voff, int row0)
    {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const f32x4b t = __builtin_bit_cast(f32x4b, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (row0 + 8 * p) * D * 4, 0));
            v[p] = make_float4(t.x, t.y, t.z, t.w);
        }
    }
    __device__ __forceinline__ void load(const float *__restrict__ base, int row0)
    {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int id = threadIdx.x + 256 * p;
            v[p] = ld4(base + (size_t)(row0 + (id >> 5)) * D + (id & 31) * 4);
        }
    }
    __device__ __forceinline__ void store(float *__restrict__ lds) const
    {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int id = threadIdx.x + 256 * p;
            *reinterpret_cast<float4 *>(lds + (id >> 5) * LDSW + (id & 31) * 4) = v[p];
        }
    }
};

template <bool QREG, bool DBUF, int SORD, int SKIP = 0, bool BLD = false>
__global__ __launch_bounds__(256, 2) void msf(const float *__restrict__ Q, const float *__restrict__ X,
                                              const float *__restrict__ bw, int N, float *__restrict__ KT,
                                              float *__restrict__ out, float *__restrict__ rsum_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_x0 = smem;                                   // X tile (buffer 0)
    float *s_x1 = smem + KB * LDSW;                       // DBUF: buffer 1; else: s_q
    float *s_rs = smem + 2 * KB * LDSW;                   // 2 * QB
    const int b = blockIdx.y, q0 = blockIdx.x * QB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int qg = wave & 1, kh = wave >> 1;
    const float *Qb = Q + (size_t)b * N * D, *Xb = X + (size_t)b * N * D;
    const float bwv = bw[b], rcp_b2 = 1.0f / (bwv * bwv);
    const int qrow = qg * 32 + li, gq = q0 + qrow;
    float *KTb = KT + (size_t)b * N * N;

    const __amdgpu_buffer_rsrc_t kt_rsrc = __builtin_amdgcn_make_buffer_rsrc(KTb, 0, N * N * 4, 0x00020000);
    const int kt_voff = ((kh * 32 + 4 * lh) * N + gq) * 4;
    float4 qf[QREG ? 16 : 1];
    Tile64 t;
    if (QREG) {
#pragma unroll
        for (int g = 0; g < 16; ++g) qf[g] = ld4(Qb + (size_t)gq * D + g * 8 + lh * 4);
    }
    if (!DBUF) { t.load(Qb, q0); t.store(s_x1); }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Xb), 0, N * D * 4, 0x00020000);
    const int x_voff = ((threadIdx.x >> 5) * D + (threadIdx.x & 31) * 4) * 4;
    if (BLD) t.loadb(x_rsrc, x_voff, 0); else t.load(Xb, 0);
    if (DBUF) { t.store(s_x0); if (KB < N) { if (BLD) t.loadb(x_rsrc, x_voff, KB); else t.load(Xb, KB); } __syncthreads(); }

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float rsum = 0.f;
    float pprev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pprev[r] = 0.f;

    for (int k0 = 0, it = 0; k0 < N; k0 += KB, ++it) {
        float *cur = DBUF ? ((it & 1) ? s_x1 : s_x0) : s_x0;
        if (!DBUF) {
            __syncthreads();
            if (!(SKIP & 8) || k0 == 0) t.store(s_x0);
            __syncthreads();
        } else if (k0 + KB < N) {
            t.store((it & 1) ? s_x0 : s_x1);   // tile it+1 (loaded during the previous step) into the other buffer
        }
        if (SORD == 0 && k0 > 0 && !(SKIP & 2)) {
            const int kb = k0 - KB + kh * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_nontemporal_store(pprev[r], KTb + (size_t)(kb + (r & 3) + 8 * (r >> 2)) * N + gq);
        }
        const int nxt = k0 + (DBUF ? 2 : 1) * KB;
        if (nxt < N && !(SKIP & 1)) { if (BLD) t.loadb(x_rsrc, x_voff, nxt); else t.load(Xb, nxt); }
        if ((SORD == 3 || SORD == 4) && k0 > 0 && !(SKIP & 2)) {
            const int sbase = (k0 - KB) * N * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pprev[r]), kt_rsrc, kt_voff,
                                                      sbase + ((r & 3) + 8 * (r >> 2)) * N * 4, SORD == 4 ? 2 : 0);
        }
        if (SORD == 2 && k0 > 0 && !(SKIP & 2)) {
            // tiled layout: tile (key block of 32, query block of 32) = 4 KiB; inside: [g 0..3][lane 0..63][4 floats]
            const int kt = (k0 - KB) / 32 + kh, qt = (q0 >> 5) + qg;
            float *tp = KTb + ((size_t)kt * (N / 32) + qt) * 1024 + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {pprev[4 * g], pprev[4 * g + 1], pprev[4 * g + 2], pprev[4 * g + 3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(tp + g * 256));
            }
        }
        if (SORD == 5 && k0 > 0 && !(SKIP & 2)) {
            const int sbase = (k0 - KB) * N * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pprev[r]), kt_rsrc, kt_voff,
                                                      sbase + ((r & 3) + 8 * (r >> 2)) * N * 4, 2);
        }
        if (SORD == 1 && k0 > 0 && !(SKIP & 2)) {
            const int kb = k0 - KB + kh * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_nontemporal_store(pprev[r], KTb + (size_t)(kb + (r & 3) + 8 * (r >> 2)) * N + gq);
        }
        const float *xa = cur + (kh * 32 + li) * LDSW + lh * 4;
        const float *qb = s_x1 + qrow * LDSW + lh * 4;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4 *>(xa + g * 8);
            const float4 bq = QREG ? qf[g] : *reinterpret_cast<const float4 *>(qb + g * 8);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, sacc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dist = 2.0f - 2.0f * sacc[r];
            float e = (-dist * rcp_b2) * 0.5f;
            e = fminf(fmaxf(e, -13.0f), 75.0f);
            const float p = (SKIP & 4) ? sacc[r] * rcp_b2 : __expf(e);
            rsum += p;
            pprev[r] = p;
            sacc[r] = p;
        }
        const float *xs = cur + (kh * 32 + 4 * lh) * LDSW + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *row = xs + ((r & 3) + 8 * (r >> 2)) * LDSW;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], row[32 * d], oacc[d], 0, 0, 0);
        }
        if (DBUF) __syncthreads();
    }
    if (SORD == 3 || SORD == 4 || SORD == 5) {
        const int sbase = (N - KB) * N * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pprev[r]), kt_rsrc, kt_voff,
                                                  sbase + ((r & 3) + 8 * (r >> 2)) * N * 4, SORD == 4 ? 2 : 0);
    } else if (SORD != 2) {
        const int kb = N - KB + kh * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_nontemporal_store(pprev[r], KTb + (size_t)(kb + (r & 3) + 8 * (r >> 2)) * N + gq);
    } else {
        const int kt = (N - KB) / 32 + kh, qt = (q0 >> 5) + qg;
        float *tp = KTb + ((size_t)kt * (N / 32) + qt) * 1024 + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {pprev[4 * g], pprev[4 * g + 1], pprev[4 * g + 2], pprev[4 * g + 3]};
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(tp + g * 256));
        }
    }
    __syncthreads();
    float *s_part = s_x0;
    rsum += __shfl_xor(rsum, 32, 64);
    if (lh == 0) s_rs[kh * QB + qrow] = rsum;
    if (kh == 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s_part[(qg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + 32 * d + li] = oacc[d][r];
    }
    __syncthreads();
    if (kh == 1) return;
    float *outb = out + (size_t)b * N * D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = qg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, gr = q0 + qr;
        const float rs = s_rs[qr] + s_rs[QB + qr], dinv = 1.0f / rs;
        float nv[4], ss = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float z = DBUF ? Qb[(size_t)gr * D + 32 * d + li] : s_x1[qr * LDSW + 32 * d + li];
            const float m = (oacc[d][r] + s_part[qr * D + 32 * d + li]) * dinv - z;
            nv[d] = z + m;
            ss += nv[d] * nv[d];
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float nrm = sqrtf(ss);
#pragma unroll
        for (int d = 0; d < 4; ++d) outb[(size_t)gr * D + 32 * d + li] = nv[d] / nrm;
        if (li == 0) rsum_out[(size_t)b * N + gr] = rs;
    }
}

template <bool QREG, bool DBUF, int SORD, int SKIP = 0, bool BLD = false>
static void run(const char *name, const float *Z, const float *X, const float *bw, int B, int N, float *KT, float *out,
                float *rs, std::vector<float> &ref_out, std::vector<float> &ref_kt)
{
    const size_t lds = (2 * KB * LDSW + 2 * QB) * sizeof(float);
    (void)hipFuncSetAttribute((const void *)msf<QREG, DBUF, SORD, SKIP, BLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((msf<QREG, DBUF, SORD, SKIP, BLD>), dim3(N / QB, B), dim3(256), lds, 0, Z, X, bw, N, KT, out, rs);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((msf<QREG, DBUF, SORD, SKIP, BLD>), dim3(N / QB, B), dim3(256), lds, 0, Z, X, bw, N, KT, out, rs);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double us = 1e3 * ms / reps;
    std::vector<float> ho((size_t)B * N * D), hk((size_t)N * N);
    (void)hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hk.data(), KT + (size_t)(B - 1) * N * N, hk.size() * 4, hipMemcpyDeviceToHost);
    double dmax = 0, kmax = 0;
    if (ref_out.empty()) { ref_out = ho; ref_kt = hk; }
    for (size_t i = 0; i < ho.size(); ++i) dmax = fmax(dmax, fabs((double)ho[i] - ref_out[i]));
    for (size_t i = 0; i < hk.size(); ++i) kmax = fmax(kmax, fabs((double)hk[i] - ref_kt[i]));
    hipError_t err = hipGetLastError();
    printf("%-46s %7.1f us  %6.1f TF/s   max|d out| %.1e  max|d K^T| %.1e  %s\n", name, us, 4.0 * B * N * (double)N * D / us / 1e6,
           dmax, kmax, err == hipSuccess ? "" : hipGetErrorString(err));
}

int main()
{
    const int B = 24, N = 2048;
    std::vector<float> hx((size_t)B * N * D);
    srand(1);
    for (size_t r = 0; r < (size_t)B * N; ++r) {
        double ss = 0;
        for (int d = 0; d < D; ++d) { float v = (float)rand() / RAND_MAX - 0.5f; hx[r * D + d] = v; ss += (double)v * v; }
        for (int d = 0; d < D; ++d) hx[r * D + d] = (float)(hx[r * D + d] / sqrt(ss));
    }
    float *X, *KT, *out, *rs, *bw;
    (void)hipMalloc(&X, hx.size() * 4); (void)hipMalloc(&KT, (size_t)B * N * N * 4); (void)hipMalloc(&out, hx.size() * 4);
    (void)hipMalloc(&rs, (size_t)B * N * 4); (void)hipMalloc(&bw, B * 4);
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hb(B, 0.6f);
    (void)hipMemcpy(bw, hb.data(), B * 4, hipMemcpyHostToDevice);
    std::vector<float> ro, rk;
    run<false, false, 0>("V0 production structure", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, false, 0>("V1 Q fragments in registers", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, true, 0>("V2 V1 + double-buffered X tile, 1 barrier", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, true, 1>("V3 V2 + K^T stores after the tile loads", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, true, 1>("V4 V3 without QREG (needs s_q: invalid)", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, false, 1>("V5 V1 + stores after loads", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 2>("V6 V0 + tiled K^T layout (4 x dwordx4 stores)", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, true, 2>("V7 V2 + tiled K^T layout", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 3>("V8 V0 + buffer stores (scalar offsets)", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 4>("V9 V8 with nt", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, true, 4>("V10 V2 + nt buffer stores", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 4, 0, true>("V11 V9 + buffer tile loads", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, true, 4, 0, true>("V12 V10 + buffer tile loads", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, false, 4, 0, true>("V13 QREG + buffer loads/stores", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, false, 5, 0, true>("V14 V13 with the stores AFTER the tile loads", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, false, 4, 2, true>("T  V13 without stores", X, X, bw, B, N, KT, out, rs, ro, rk);
    // timing-only eliminations on V0 (results are wrong by construction)
    run<false, false, 0, 2>("T  no K^T stores", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 0, 1>("T  no tile loads", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 0, 3>("T  no tile loads, no stores", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 0, 11>("T  no loads/stores/LDS tile writes", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 0, 15>("T  ... and no exp", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<false, false, 0, 4>("T  no exp only", X, X, bw, B, N, KT, out, rs, ro, rk);
    run<true, true, 0, 11>("T  DBUF+QREG no loads/stores/LDS writes", X, X, bw, B, N, KT, out, rs, ro, rk);
    return 0;
}
