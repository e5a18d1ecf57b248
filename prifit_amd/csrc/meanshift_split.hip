// LABELLED EXPERIMENT, default off (fit_ops.MS_SPLIT / bench.py --ms-split): the two products of a mean-shift update
// (src/mean_shift.py:65 S = Z X^T and :73 O = K X) on the 16-bit matrix pipe with ERROR-COMPENSATED operands.
//
// fp32 MFMA runs at 1/16 of the bf16 / fp16 rate on gfx950, and the fused fp32 kernel (meanshift_fused.hip) sits at 0.8
// of that peak.  Here every fp32 operand x is cut into NP 16-bit planes, x = x0 + x1 (+ x2) with x0 = rn16(x),
// x1 = rn16(x - x0), x2 = rn16(x - x0 - x1) (the differences are exact in fp32), and a product a.b is the sum of the
// plane products whose weight is above the target error, accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16}:
//     bf16x3  NP = 2, a0 b0 + a0 b1 + a1 b0                       (operands to 2^-16: NOT fp32 grade)
//     bf16x6  NP = 3, + a1 b1 + a0 b2 + a2 b0                     (operands to 2^-24)
//     fp16x3  NP = 2 fp16 planes (11 + 11 bits) of 16 x and 1024 K (powers of two, undone exactly afterwards: the low
//             planes stay out of fp16's subnormal range), a0 b0 + a0 b1 + a1 b0: operands to ~2^-22 with half of bf16x6's
//             matrix work
// The exponent, the clamp and the row sums stay fp32 on the vector pipe, exactly the fused kernel's expressions.
//
// Data flow (D = 128, N % 256 == 0): a preparation launch cuts the dictionary X once per mean-shift call (it is the
// same in all iterations, :65) into planes, in the two images the matrix operands want -- rows [key][d] for S, and
// [dim][key slot] with the keys of a 32-key tile permuted into the order in which an accumulator tile presents them as
// an operand (slot 16 s + 8 h + j = key 16 s + 4 h + (j & 3) + 8 (j >> 2)) for O.  One workgroup = 8 waves x 32 queries;
// its Z rows live in registers as B fragments for the whole launch; per 32-key step the pre-cut tile (16 KB per plane)
// goes global -> registers -> LDS (two stages, one barrier per step);  S^T = X_tile Z^T leaves the keys on the accumulator
// rows and the queries on the lanes, K = exp(..) is cut in registers and is the A operand of O += K X_tile as it stands.
// Outputs O [B,N,128] and rowsum [B,N]; the normalisation is prifit_meanshift_update_fwd, as for the GEMM chain.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int D = 128;
constexpr int KT = 32;              // keys per step
constexpr int QB = 256;             // queries per workgroup (8 waves x 32)
constexpr int NTH = 512;
constexpr int XB_ROW = 2 * D + 16;  // bytes of a key row in LDS (padded: conflict-free ds_read_b128 over 16 lanes)
constexpr int XT_ROW = 2 * KT + 16; // bytes of a dim row in LDS
constexpr int XB_SZ = KT * XB_ROW, XT_SZ = D * XT_ROW, PLANE_SZ = XB_SZ + XT_SZ;
constexpr int PLANE_G = 2 * KT * D * 2;   // bytes of one plane of one 32-key tile in the workspace (both images, unpadded)

struct BF16 {
    typedef __bf16 E;
    typedef bf16x8 V;
    static constexpr float SX = 1.f, SP = 1.f;
    static __device__ __forceinline__ f32x16 mma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct FP16 {
    typedef _Float16 E;
    typedef f16x8 V;
    static constexpr float SX = 16.f, SP = 1024.f;
    static __device__ __forceinline__ f32x16 mma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <class TR, int NP>
__device__ __forceinline__ void cut(float v, typename TR::E (&e)[NP])
{
    float r = v;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        e[p] = (typename TR::E)r;
        r -= (float)e[p];
    }
}

// key of slot s (0..31) of a 32-key tile in the [dim][slot] image
__device__ __forceinline__ int slot_key(int s) { const int j = s & 7; return (s & 16) + 4 * ((s >> 3) & 1) + (j & 3) + 8 * (j >> 2); }

// X [B,N,128] fp32 -> ws: per (shape, 32-key tile) NP planes of { rows image 32 x 128 | slot image 128 x 32 }, 16-byte chunks
template <class TR, int NP>
__global__ __launch_bounds__(NTH) void ms_split_prep_kernel(const float *__restrict__ X, int N, unsigned char *__restrict__ ws)
{
    typedef typename TR::E E;
    typedef typename TR::V V;
    const int b = blockIdx.y, kb = blockIdx.x, t = threadIdx.x;
    const float *Xt = X + ((size_t)b * N + (size_t)kb * KT) * D;
    unsigned char *dst = ws + ((size_t)b * (N / KT) + kb) * (size_t)(NP * PLANE_G);
    V rowv[NP], slotv[NP];
    {
        const int key = t >> 4, part = t & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            E e[NP];
            cut<TR, NP>(Xt[key * D + 8 * part + j] * TR::SX, e);
#pragma unroll
            for (int p = 0; p < NP; ++p) rowv[p][j] = e[p];
        }
    }
    {
        const int dim = t >> 2, part = t & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            E e[NP];
            cut<TR, NP>(Xt[slot_key(8 * part + j) * D + dim] * TR::SX, e);
#pragma unroll
            for (int p = 0; p < NP; ++p) slotv[p][j] = e[p];
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        *reinterpret_cast<V *>(dst + (size_t)p * PLANE_G + (size_t)t * 16) = rowv[p];
        *reinterpret_cast<V *>(dst + (size_t)p * PLANE_G + PLANE_G / 2 + (size_t)t * 16) = slotv[p];
    }
}

template <class TR, int NP>
__global__ __launch_bounds__(NTH, 1) void ms_split_fwd_kernel(const float *__restrict__ Z, const unsigned char *__restrict__ ws,
                                                               const float *__restrict__ bw, int B, int N,
                                                               float *__restrict__ O, float *__restrict__ rowsum)
{
    typedef typename TR::E E;
    typedef typename TR::V V;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * NP * PLANE_SZ];

    // the query blocks of a shape run on ONE XCD (they stream the same 16 KB x NP x N / 32 of cut dictionary: one L2)
    const int qblocks = N / QB;
    int b, qb;
    {
        const int L = blockIdx.x;
        if ((B & 7) == 0) {
            const int xcd = L & 7, j = L >> 3;
            qb = j % qblocks;
            b = xcd + 8 * (j / qblocks);
        } else {
            b = L / qblocks; qb = L - b * qblocks;
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = qb * QB + 32 * wave;

    const float bwv = bw[b];
    const float c_e2 = (1.0f / (bwv * bwv)) * 1.44269504088896341f;   // log2(e) / b^2, as in meanshift_fused.hip
    constexpr float US = 1.f / (TR::SX * TR::SX), UO = 1.f / (TR::SX * TR::SP);

    // this wave's queries as B fragments: lane (query li, half lh) holds d = 16 kk + 8 lh + 0..7 of every plane
    V zf[NP][8];
    {
        const float *zrow = Z + ((size_t)b * N + q0 + li) * D + 8 * lh;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 a = *reinterpret_cast<const float4 *>(zrow + 16 * kk);
            const float4 c = *reinterpret_cast<const float4 *>(zrow + 16 * kk + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                E e[NP];
                cut<TR, NP>(v[j] * TR::SX, e);
#pragma unroll
                for (int p = 0; p < NP; ++p) zf[p][kk][j] = e[p];
            }
        }
    }

    const int steps = N / KT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(ws + (size_t)b * steps * (size_t)(NP * PLANE_G)), 0, steps * NP * PLANE_G, 0x00020000);
    const int g_voff = threadIdx.x * 16;
    const int xb_off = (threadIdx.x >> 4) * XB_ROW + (threadIdx.x & 15) * 16;
    const int xt_off = XB_SZ + (threadIdx.x >> 2) * XT_ROW + (threadIdx.x & 3) * 16;
    u32x4 pf[2 * NP];
    auto load_tile = [&](int step) {
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i)
            pf[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, g_voff, step * (NP * PLANE_G) + i * (PLANE_G / 2), 0));
    };
    auto store_tile = [&](unsigned char *stage) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<u32x4 *>(stage + p * PLANE_SZ + xb_off) = pf[2 * p];
            *reinterpret_cast<u32x4 *>(stage + p * PLANE_SZ + xt_off) = pf[2 * p + 1];
        }
    };

    f32x16 oacc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;
    float rsum = 0.f;

    load_tile(0);
    store_tile(lds);
    __syncthreads();
    for (int step = 0; step < steps; ++step) {
        const unsigned char *st = lds + (step & 1) * (NP * PLANE_SZ);
        const int nstep = step + 1 < steps ? step + 1 : step;
        load_tile(nstep);
        __builtin_amdgcn_sched_barrier(0);

        // ---- S^T tile = X_tile Z_q^T: keys on the accumulator rows, queries on the lanes; small terms first
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const unsigned char *xa_p = st + li * XB_ROW + 16 * lh;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            V xa[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) xa[p] = *reinterpret_cast<const V *>(xa_p + p * PLANE_SZ + 32 * kk);
            if (NP == 3) {
                sacc = TR::mma(xa[0], zf[NP - 1][kk], sacc);
                sacc = TR::mma(xa[NP - 1], zf[0][kk], sacc);
                sacc = TR::mma(xa[1], zf[1][kk], sacc);
            }
            sacc = TR::mma(xa[0], zf[1][kk], sacc);
            sacc = TR::mma(xa[1], zf[0][kk], sacc);
            sacc = TR::mma(xa[0], zf[0][kk], sacc);
        }

        // ---- K = exp(clamp((S - 1) / b^2, -13, 75)) (src/mean_shift.py:65-68, src/guard.py:6-11), cut into planes:
        // registers 8 s .. 8 s + 7 are the A fragment of k-step s of the second product
        V pa[NP][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = fminf(fmaxf(fmaf(sacc[r] * US, c_e2, -c_e2), -13.0f * 1.44269504088896341f), 75.0f * 1.44269504088896341f);
            const float pv = __builtin_amdgcn_exp2f(t);
            rsum += pv;
            E e[NP];
            cut<TR, NP>(pv * TR::SP, e);
#pragma unroll
            for (int p = 0; p < NP; ++p) pa[p][r >> 3][r & 7] = e[p];
        }

        // ---- O_q += K X_tile: lanes = dims of block n, accumulator rows = queries
        const unsigned char *xb_p = st + XB_SZ + li * XT_ROW + 16 * lh;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                V xb[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) xb[p] = *reinterpret_cast<const V *>(xb_p + p * PLANE_SZ + 32 * n * XT_ROW + 32 * s);
                if (NP == 3) {
                    oacc[n] = TR::mma(pa[0][s], xb[NP - 1], oacc[n]);
                    oacc[n] = TR::mma(pa[NP - 1][s], xb[0], oacc[n]);
                    oacc[n] = TR::mma(pa[1][s], xb[1], oacc[n]);
                }
                oacc[n] = TR::mma(pa[0][s], xb[1], oacc[n]);
                oacc[n] = TR::mma(pa[1][s], xb[0], oacc[n]);
                oacc[n] = TR::mma(pa[0][s], xb[0], oacc[n]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tile(lds + ((step + 1) & 1) * (NP * PLANE_SZ));
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31 (dim), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query)
    float *Ob = O + ((size_t)b * N + q0) * D;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) Ob[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * D + 32 * n + li] = oacc[n][r] * UO;
    rsum += __shfl_xor(rsum, 32, 64);
    if (lh == 0) rowsum[(size_t)b * N + q0 + li] = rsum;
}

template <class TR, int NP>
int launch_prep(const float *X, int B, int N, void *ws, hipStream_t st)
{
    hipLaunchKernelGGL((ms_split_prep_kernel<TR, NP>), dim3(N / KT, B), dim3(NTH), 0, st, X, N, (unsigned char *)ws);
    return prifit_check_launch();
}
template <class TR, int NP>
int launch_fwd(const float *Z, const void *ws, const float *bw, int B, int N, float *O, float *rowsum, hipStream_t st)
{
    hipLaunchKernelGGL((ms_split_fwd_kernel<TR, NP>), dim3(B * (N / QB)), dim3(NTH), 0, st, Z, (const unsigned char *)ws, bw, B,
                       N, O, rowsum);
    return prifit_check_launch();
}
int planes_of(int mode) { return mode == PRIFIT_SPLIT_BF16X6 ? 3 : 2; }
}  // namespace

extern "C" {

int prifit_meanshift_split_supported(int N, int D_, int mode)
{
    return D_ == D && N > 0 && N % QB == 0 && N <= 16384 &&
           (mode == PRIFIT_SPLIT_BF16X3 || mode == PRIFIT_SPLIT_BF16X6 || mode == PRIFIT_SPLIT_FP16X3);
}

long long prifit_meanshift_split_workspace(int B, int N, int D_, int mode)
{
    if (!prifit_meanshift_split_supported(N, D_, mode) || B <= 0) return 0;
    return (long long)B * (N / KT) * planes_of(mode) * PLANE_G;
}

int prifit_meanshift_split_prep(const float *X, int B, int N, int D_, int mode, void *workspace, void *stream)
{
    if (!X || !workspace || B <= 0 || !prifit_meanshift_split_supported(N, D_, mode) || ((uintptr_t)workspace & 15))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if (mode == PRIFIT_SPLIT_BF16X3) return launch_prep<BF16, 2>(X, B, N, workspace, st);
    if (mode == PRIFIT_SPLIT_BF16X6) return launch_prep<BF16, 3>(X, B, N, workspace, st);
    return launch_prep<FP16, 2>(X, B, N, workspace, st);
}

int prifit_meanshift_split_fwd(const float *Z, const void *workspace, const float *bw, int B, int N, int D_, int mode,
                               float *O, float *rowsum, void *stream)
{
    if (!Z || !workspace || !bw || !O || !rowsum || B <= 0 || !prifit_meanshift_split_supported(N, D_, mode) ||
        ((uintptr_t)workspace & 15) || ((uintptr_t)Z & 15))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if (mode == PRIFIT_SPLIT_BF16X3) return launch_fwd<BF16, 2>(Z, workspace, bw, B, N, O, rowsum, st);
    if (mode == PRIFIT_SPLIT_BF16X6) return launch_fwd<BF16, 3>(Z, workspace, bw, B, N, O, rowsum, st);
    return launch_fwd<FP16, 2>(Z, workspace, bw, B, N, O, rowsum, st);
}

}  // extern "C"
