// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// One templated kernel serves every dense contraction of the hot path:
//   NT  C[M,N] = A[M,K] . B[N,K]^T   shared-MLP forward (1x1 conv), S = Z X^T of mean-shift
//   NN  C[M,N] = A[M,K] . B[K,N]     dA = dY . W, (K X) of mean-shift
//   TN  C[M,N] = A[K,M]^T . B[K,N]   dW = dY^T . A (split over the reduction), dX of mean-shift
// with fused prologues (train-mode BatchNorm + ReLU of the producing layer applied while the
// operand is staged: "normalise on load") and epilogues (bias, per-column sum / sum-of-squares
// partials for the next BatchNorm, chord-distance and mean-shift kernel transforms).
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; each wave owns WM x WN = (32*TM) x (32*TN)
// accumulators (TM*TN f32x16).  Operand tiles are staged global -> registers -> LDS with the next
// tile's global loads in flight during the MFMAs of the current one.  LDS rows are padded by one
// 16-byte slot so that the ds_read_b128 fragment reads are bank-conflict free on the 64-bank LDS.
// Block ids are remapped so that the N-tiles of one M-panel run on the same XCD (shared L2).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { LAY_NT = 0, LAY_NN = 1, LAY_TN = 2 };
enum { EPI_NONE = 0, EPI_CHORD = 1, EPI_MSKERNEL = 2, EPI_MSBWD = 3, EPI_BNRED = 4 };

struct GemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    long long lda, ldb, ldc;
    long long sA, sB, sC;  // batch strides (elements)
    int batch, splitk;     // gridDim.z = batch * splitk
    const float *a_scale, *a_shift;  // prologue on A: a' = max(a*scale[c]+shift[c], 0), c = A's contiguous index
    const float *b_scale, *b_shift;  // same for B
    const float *bias;               // [N] (+ z * bias_stride), added to every row
    long long bias_stride;
    float *stats;                    // [tilesM][2][N] per-column partial (sum, sum of squares) of C
    int epi;
    const float *epi_batch_scalar;   // EPI_MSKERNEL / EPI_MSBWD: bandwidth b[z]
    const float *aux;                // EPI_MSBWD: the forward kernel matrix, indexed like C
    const float *row_add;            // EPI_MSBWD: per-row additive term [batch][M] (may be NULL)
    float *a_rowsum;                 // NN/NT only: sum over k of the staged A rows -> [batch][M] (may be NULL)
    long long ldaux, sAux;
    // EPI_BNRED (dA products): aux = the previous layer's pre-activations Yp (indexed like C); C is stored unchanged
    // and `stats` receives, instead of (sum, sum of squares) of C, the BatchNorm-backward partials
    // m1 = sum(C * mask), m2 = sum(C * mask * yhat), mask = (Yp*red_scale+red_shift > 0), yhat = (Yp-red_mean)*red_invstd
    const float *red_scale, *red_shift, *red_mean, *red_invstd;
    // Dual source: k-tiles >= kswitch read A + dA2 / B + dB2 (element offsets that already absorb the k shift), so that
    // C = A1 B1 + A2 B2 runs as ONE product over K = K1 + K2 when the two pairs share their leading dimensions
    // (mean-shift backward: dX += gS^T Z + K^T gO, one epilogue instead of two).  kswitch = 0: off.
    long long dA2, dB2;
    int kswitch;
    int accumulate;                  // 1: C += result (atomics when split-K); 0: store
    int vecA, vecB;                  // 16-byte loads legal for the operand
    int ntiles;                      // gemm_pers_kernel: output tiles, walked with a grid stride
    float *cand;                     // gemm_pers_kernel<.., PMAX>: [M / 32][4][N] pool candidates
    BnTail tail;                     // the column sums `stats` would receive, finalized by this launch instead (common.h)
};

#ifndef GEMM_W8
#define GEMM_W8 6
#endif
constexpr int BK = 32;
constexpr int PAD = 4;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Stage one operand tile (ROWS x BK, logical [row][k]) from global into registers.
//  KC = true : stored [R][K] (k contiguous);  KC = false: stored [K][R] (row index contiguous).
//  VEC: 16-byte loads (extents / strides multiples of 4 floats).  AFF: apply max(x*scale[c]+shift[c], 0).
// Everything that does not depend on the k-tile (row clamps, row predicates, base pointers, the prologue
// coefficients when the channel is the row) is computed once in init(); load(kt) is pointer + kt*step.
// Loads are branch-free: out-of-range lanes read a clamped (valid) address and the value is zeroed
// afterwards, so all loads of a tile are in flight together (hipcc serialises predicated loads with a
// vmcnt(0) each).
template <int ROWS, bool KC, bool VEC, bool AFF, int NTH>
struct TileLoader {
    static constexpr int NV = ROWS * BK / 4 / NTH;  // float4 per thread
    static_assert(NV * NTH * 4 == ROWS * BK, "tile must divide evenly over the threads");
    float4 v[NV];
    const float *ptr[NV];   // element 0 of this thread's float4 in k-tile 0 (row / column clamped into range)
    int klo[NV];            // k index of element 0 inside the tile
    int c0[NV];             // !KC: channel (= global row index) of element 0
    unsigned okmask[NV];    // per element: the row / column exists
    long long step;         // pointer increment per k-tile
    int K, R, cclamp;

    __device__ __forceinline__ void init(const float *__restrict__ base, long long ld, int r0, int R_, int K_)
    {
        K = K_; R = R_;
        step = KC ? (long long)BK : (long long)BK * ld;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (KC) {
                const int gr = r0 + id / (BK / 4);
                klo[p] = (id % (BK / 4)) * 4;
                okmask[p] = gr < R ? 0xFu : 0u;
                // K < BK: columns beyond K are never valid; keep the pointer inside the row
                ptr[p] = base + (long long)(gr < R ? gr : 0) * ld + (klo[p] < K ? klo[p] : 0);
                c0[p] = 0;
            } else {
                klo[p] = id / (ROWS / 4);
                const int gr = r0 + (id % (ROWS / 4)) * 4;
                unsigned m = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) m |= (gr + j < R) ? (1u << j) : 0u;
                okmask[p] = m;
                c0[p] = m ? gr : 0;
                ptr[p] = base + (long long)klo[p] * ld + c0[p];
            }
        }
    }

    // load(): ONLY the global loads (clamped addresses, nothing that consumes the data) -- the prologue and the zeroing of
    // out-of-range elements happen in store(), when the tile is written to LDS after the MFMAs of the current tile.  With
    // the select right behind the load the compiler put an s_waitcnt vmcnt(0) after every load, i.e. the "next tile in
    // flight during the MFMAs" never was in flight.
    int kload;
    __device__ __forceinline__ void load(int k0, const float *__restrict__, const float *__restrict__)
    {
        kload = k0;
        const int kt = k0 / BK;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int gk = k0 + klo[p];
            const bool kin = gk < K;                       // element 0 (VEC: K % 4 == 0 covers all four)
            const float *src = ptr[p] + (kin ? (long long)kt * step : 0);
            if (!KC && !kin && klo[p] >= K) src = ptr[p] - (long long)klo[p] * (step / BK);  // K < BK: stay in row 0
            if (VEC) {
                v[p] = ld4(src);
            } else {
                // scalar path: element j may lie beyond the row / the K extent -> read element 0 again instead
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = KC ? (okmask[p] && (gk + j) < K) : (kin && ((okmask[p] >> j) & 1u));
                    e[j] = src[ok ? j : 0];
                }
                v[p] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }

    // prologue + masks of the tile loaded last (kload), applied in registers (v keeps the transformed values: the A-row
    // sums of the caller read them)
    __device__ __forceinline__ void finish(const float *__restrict__ scale, const float *__restrict__ shift)
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int gk = kload + klo[p];
            const bool kin = gk < K;
            float4 x = v[p];
            if (VEC) {
                if (AFF) {
                    const int cc = KC ? (kin ? gk : 0) : c0[p];
                    const float4 s = ld4(scale + cc), t = ld4(shift + cc);
                    x.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
                    x.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
                }
                if (!(kin && okmask[p])) x = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = KC ? (okmask[p] && (gk + j) < K) : (kin && ((okmask[p] >> j) & 1u));
                    float val = ok ? e[j] : 0.f;
                    if (AFF && ok) {
                        const int cc = KC ? gk + j : c0[p] + j;
                        val = fmaxf(fmaf(val, scale[cc], shift[cc]), 0.f);
                    }
                    e[j] = val;
                }
                x = make_float4(e[0], e[1], e[2], e[3]);
            }
            v[p] = x;
        }
    }

    __device__ __forceinline__ void shift(long long delta)
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) ptr[p] += delta;
    }

    __device__ __forceinline__ void store(float *__restrict__ lds) const
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (KC) {
                const int row = id / (BK / 4), kk = (id % (BK / 4)) * 4;
                *reinterpret_cast<float4 *>(lds + row * (BK + PAD) + kk) = v[p];
            } else {
                const int kk = id / (ROWS / 4), row = (id % (ROWS / 4)) * 4;
                *reinterpret_cast<float4 *>(lds + kk * (ROWS + PAD) + row) = v[p];
            }
        }
    }
};

// The same staging for operands whose extents are whole tiles (rows % ROWS == 0, K % BK == 0, 16-byte aligned rows) with
// BUFFER addressing: one constant lane offset per float4 (computed once), the k-tile as a SCALAR offset -- no 64-bit
// vector arithmetic, no predicates, no selects per k-tile (what the generic loader spends ~10 VALU per float4 on).
// The matrix (one batch item) must span < 2 GiB.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
template <int ROWS, bool KC, bool AFF, int NTH>
struct ExactLoader {
    static constexpr int NV = ROWS * BK / 4 / NTH;
    static_assert(NV * NTH * 4 == ROWS * BK, "tile must divide evenly over the threads");
    float4 v[NV];
    int voff[NV];
    int klo[NV], c0[NV];
    const float *base;
    __amdgpu_buffer_rsrc_t rs;
    int step4;      // bytes per k-tile
    int kload;

    __device__ __forceinline__ void init(const float *__restrict__ b, long long ld, int r0, int, int)
    {
        base = b;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b), 0, 0x7ffffffc, 0x00020000);
        step4 = __builtin_amdgcn_readfirstlane(KC ? BK * 4 : (int)(BK * ld * 4));
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (KC) {
                klo[p] = (id % (BK / 4)) * 4;
                c0[p] = 0;
                voff[p] = (int)(((long long)(r0 + id / (BK / 4)) * ld + klo[p]) * 4);
            } else {
                klo[p] = id / (ROWS / 4);
                c0[p] = r0 + (id % (ROWS / 4)) * 4;
                voff[p] = (int)(((long long)klo[p] * ld + c0[p]) * 4);
            }
        }
    }
    __device__ __forceinline__ void load(int k0, const float *__restrict__, const float *__restrict__)
    {
        kload = k0;
        const int soff = __builtin_amdgcn_readfirstlane((k0 / BK) * step4);
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const f32x4v t = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], soff, 0));
            v[p] = make_float4(t.x, t.y, t.z, t.w);
        }
    }
    __device__ __forceinline__ void finish(const float *__restrict__ scale, const float *__restrict__ shift)
    {
        if (!AFF) return;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int cc = KC ? kload + klo[p] : c0[p];
            const float4 s = ld4(scale + cc), t = ld4(shift + cc);
            float4 x = v[p];
            x.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
            x.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
            v[p] = x;
        }
    }
    __device__ __forceinline__ void shift(long long delta)   // second source of a dual product: a new resource
    {
        base += delta;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffc, 0x00020000);
    }
    __device__ __forceinline__ void store(float *__restrict__ lds) const
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (KC) {
                const int row = id / (BK / 4), kk = (id % (BK / 4)) * 4;
                *reinterpret_cast<float4 *>(lds + row * (BK + PAD) + kk) = v[p];
            } else {
                const int kk = id / (ROWS / 4), row = (id % (ROWS / 4)) * 4;
                *reinterpret_cast<float4 *>(lds + kk * (ROWS + PAD) + row) = v[p];
            }
        }
    }
};

// Buffer-addressed staging for ANY extents (16-byte rows): what ExactLoader does for whole tiles, plus rows / columns
// beyond the operand (lane offset that fails the bounds check: zeros) and a partial last k-tile (one compare per load,
// in that k-tile only).  The matrix (one batch item) must span < 2 GiB.
template <int ROWS, bool KC, bool AFF, int NTH>
struct BufLoader {
    static constexpr int NV = ROWS * BK / 4 / NTH;
    static_assert(NV * NTH * 4 == ROWS * BK, "tile must divide evenly over the threads");
    static constexpr int OOBV = 0x7fffffff;
    float4 v[NV];
    int voff[NV];       // byte offset of this thread's float4 in k-tile 0, OOBV when its row / column group does not exist
    int klo[NV], c0[NV];
    __amdgpu_buffer_rsrc_t rs;
    const float *base;
    int step4, K, kload;

    __device__ __forceinline__ void init(const float *__restrict__ b, long long ld, int r0, int R, int K_)
    {
        K = K_;
        base = b;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b), 0, 0x7ffffffc, 0x00020000);
        step4 = KC ? BK * 4 : (int)(BK * ld * 4);
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (KC) {
                const int row = r0 + id / (BK / 4);
                klo[p] = (id % (BK / 4)) * 4;
                c0[p] = 0;
                voff[p] = row < R ? (int)(((long long)row * ld + klo[p]) * 4) : OOBV;
            } else {
                klo[p] = id / (ROWS / 4);
                c0[p] = r0 + (id % (ROWS / 4)) * 4;   // R % 4 == 0: the four rows exist together
                voff[p] = c0[p] < R ? (int)(((long long)klo[p] * ld + c0[p]) * 4) : OOBV;
                if (c0[p] >= R) c0[p] = 0;
            }
        }
    }
    __device__ __forceinline__ void load(int k0, const float *__restrict__, const float *__restrict__) { load(k0); }
    __device__ __forceinline__ void shift(long long delta)   // second source of a dual product: a new resource
    {
        base += delta;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7ffffffc, 0x00020000);
    }
    __device__ __forceinline__ void load(int k0)
    {
        kload = k0;
        const int soff = (k0 / BK) * step4;
        if (k0 + BK <= K) {   // block-uniform
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const f32x4v t = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], soff, 0));
                v[p] = make_float4(t.x, t.y, t.z, t.w);
            }
        } else {              // last, partial k-tile (K % 4 == 0: a float4 is inside or outside as a whole)
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int vo = k0 + klo[p] < K ? voff[p] : OOBV;
                const f32x4v t = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
                v[p] = make_float4(t.x, t.y, t.z, t.w);
            }
        }
    }
    __device__ __forceinline__ void finish(const float *__restrict__ scale, const float *__restrict__ shift)
    {
        if (!AFF) return;     // (elements that do not exist arrived as zeros)
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int gk = kload + klo[p];
            const bool ok = voff[p] != OOBV && gk < K;
            const int cc = KC ? (gk < K ? gk : 0) : c0[p];
            const float4 s = ld4(scale + cc), t = ld4(shift + cc);
            float4 x = v[p];
            x.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
            x.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
            v[p] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void store(float *__restrict__ lds) const
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (KC) {
                const int row = id / (BK / 4), kk = (id % (BK / 4)) * 4;
                *reinterpret_cast<float4 *>(lds + row * (BK + PAD) + kk) = v[p];
            } else {
                const int kk = id / (ROWS / 4), row = (id % (ROWS / 4)) * 4;
                *reinterpret_cast<float4 *>(lds + kk * (ROWS + PAD) + row) = v[p];
            }
        }
    }
};

// Fragment of 4 consecutive MFMA k-steps for lane (i, h): k = 8*g + 4*h + j, j = 0..3.
template <int ROWS, bool KC>
__device__ __forceinline__ float4 read_frag(const float *__restrict__ lds, int row, int g, int h)
{
    if (KC) return *reinterpret_cast<const float4 *>(lds + row * (BK + PAD) + g * 8 + h * 4);
    const float *p = lds + (g * 8 + h * 4) * (ROWS + PAD) + row;
    return make_float4(p[0], p[ROWS + PAD], p[2 * (ROWS + PAD)], p[3 * (ROWS + PAD)]);
}

// The MFMAs of one staged k-tile for a wave with a 32 x (32 TN) block: the fragments of k group g + 1 are requested before
// the 2 TN x 4 matrix instructions of group g and pinned there (left to itself the compiler reads each operand two MFMAs
// ahead of its use -- 128 cycles of cover for an LDS round trip under load).  ngk < BK / 8: a ragged last k-tile.
// Same products in the same order as the plain loop: bit-identical sums.
template <int RA, bool AKC, int RB, bool BKC, int TN>
__device__ __forceinline__ void mma_ktile(const float *__restrict__ As, const float *__restrict__ Bs, int arow, int bcol, int lh,
                                          f32x16 (&acc)[TN], int ngk = BK / 8, int nbv = TN)
{
    // nbv < TN (wave-uniform): the wave's column blocks from nbv on lie beyond N (a ragged last tile: 196 = 128 + 68 leaves
    // the fourth 32-column block of the second tile empty) -- neither read nor multiplied, their accumulators stay zero
    float4 fa[2], fb[2][TN];
    fa[0] = read_frag<RA, AKC>(As, arow, 0, lh);
#pragma unroll
    for (int b = 0; b < TN; ++b)
        if (b < nbv) fb[0][b] = read_frag<RB, BKC>(Bs, bcol + 32 * b, 0, lh);
#pragma unroll
    for (int gk = 0; gk < BK / 8; ++gk) {
        if (gk > 0 && gk >= ngk) break;   // (scalar)
        if (gk + 1 < BK / 8) {
            fa[(gk + 1) & 1] = read_frag<RA, AKC>(As, arow, gk + 1, lh);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                if (b < nbv) fb[(gk + 1) & 1][b] = read_frag<RB, BKC>(Bs, bcol + 32 * b, gk + 1, lh);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float4 a = fa[gk & 1];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            if (b >= nbv) continue;
            const float4 w = fb[gk & 1][b];
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc[b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool C, class A, class B> struct pick { typedef A type; };
template <class A, class B> struct pick<false, A, B> { typedef B type; };

// VA/VB: vector loads legal for A/B; FA/FB: BatchNorm+ReLU prologue on A/B; EX: whole tiles only (ExactLoader).
template <int BM, int BN, int WM, int WN, int LAY, bool VA, bool VB, bool FA, bool FB, bool EX = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, ((BM / WM) * (BN / WN) == 8 ? GEMM_W8 : 4)) void gemm_kernel(const GemmArgs g)
{
    constexpr int NTH = (BM / WM) * (BN / WN) * 64;
    constexpr bool A_KC = (LAY != LAY_TN);
    constexpr bool B_KC = (LAY == LAY_NT);
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert(NTH == 256 || NTH == 512, "4 or 8 waves per block");
    constexpr int SZA = A_KC ? BM * (BK + PAD) : BK * (BM + PAD);
    constexpr int SZB = B_KC ? BN * (BK + PAD) : BK * (BN + PAD);
    __shared__ __attribute__((aligned(16))) float lds[2 * (SZA + SZB)];  // two stages

    // XCD-aware bijective remap: logical tiles that share an A panel get ids that are consecutive on
    // one XCD (hardware deals consecutive workgroup ids round-robin over the 8 XCDs).
    const int tilesN = (g.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x, xcd = bid & 7, pos = bid >> 3, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    const int tile_m = L / tilesN, tile_n = L - tile_m * tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z / g.splitk, ks = blockIdx.z - z * g.splitk;

    const float *A = g.A + (long long)z * g.sA;
    const float *B = g.B + (long long)z * g.sB;
    float *C = g.C + (long long)z * g.sC;

    // K range of this split
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    typename pick<EX, BufLoader<BM, A_KC, FA, NTH>, TileLoader<BM, A_KC, VA, FA, NTH>>::type la;
    typename pick<EX, BufLoader<BN, B_KC, FB, NTH>, TileLoader<BN, B_KC, VB, FB, NTH>>::type lb;
    la.init(A, g.lda, m0, g.M, g.K);
    lb.init(B, g.ldb, n0, g.N, g.K);
    constexpr int NVA = TileLoader<BM, A_KC, VA, FA, NTH>::NV;
    float rs[NVA];
#pragma unroll
    for (int p = 0; p < NVA; ++p) rs[p] = 0.f;
    const bool want_rowsum = !EX && A_KC && g.a_rowsum != nullptr && tile_n == 0;

    // Software pipeline: tile kt is computed from LDS stage s while tile kt+1 travels global -> registers;
    // it is written to stage s^1 after the MFMAs (nobody reads s^1 any more: the barrier that ended the
    // previous iteration) and one barrier per k-tile publishes it.
    const int kswitch = g.kswitch ? g.kswitch : 0x7fffffff;  // first k-tile of the second source
    if (kt0 >= kswitch) { la.shift(g.dA2); lb.shift(g.dB2); }
    if (kt0 < kt1) {
        la.load(kt0 * BK, g.a_scale, g.a_shift);
        lb.load(kt0 * BK, g.b_scale, g.b_shift);
        la.finish(g.a_scale, g.a_shift);
        lb.finish(g.b_scale, g.b_shift);
        la.store(lds);
        lb.store(lds + SZA);
        if (want_rowsum) {
#pragma unroll
            for (int p = 0; p < NVA; ++p) rs[p] += (la.v[p].x + la.v[p].y) + (la.v[p].z + la.v[p].w);
        }
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int stage = (kt - kt0) & 1;
        const float *As = lds + stage * (SZA + SZB), *Bs = As + SZA;
        const bool more = kt + 1 < kt1;
        if (more) {
            if (kt + 1 == kswitch) { la.shift(g.dA2); lb.shift(g.dB2); }
            la.load((kt + 1) * BK, g.a_scale, g.a_shift);
            lb.load((kt + 1) * BK, g.b_scale, g.b_shift);
        }
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = read_frag<BM, A_KC>(As, wm0 + 32 * a + li, gk, lh);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = read_frag<BN, B_KC>(Bs, wn0 + 32 * b + li, gk, lh);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
        if (more) {
            float *An = lds + (stage ^ 1) * (SZA + SZB);
            la.finish(g.a_scale, g.a_shift);
            lb.finish(g.b_scale, g.b_shift);
            la.store(An);
            lb.store(An + SZA);
            if (want_rowsum) {
#pragma unroll
                for (int p = 0; p < NVA; ++p) rs[p] += (la.v[p].x + la.v[p].y) + (la.v[p].z + la.v[p].w);
            }
        }
        __syncthreads();
    }

    if (want_rowsum) {  // KC staging: float4 p of thread t belongs to row (t + 256 p) / 8; 8 lanes share a row
#pragma unroll
        for (int p = 0; p < NVA; ++p) {
            float v = rs[p];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            const int row = m0 + (threadIdx.x + NTH * p) / (BK / 4);
            if ((threadIdx.x & 7) == 0 && row < g.M) g.a_rowsum[(long long)z * g.M + row] = v;
        }
    }

    if (g.epi == 100) {  // timing-only build switch (tools/gemm_bench.py --nostore): price of the epilogue
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[a][b][r]));
        return;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float inv_b2 = 0.f;
    if (g.epi == EPI_MSKERNEL || g.epi == EPI_MSBWD) { const float bw = g.epi_batch_scalar[z]; inv_b2 = bw * bw; }
    const float rcp_b2 = inv_b2 > 0.f ? 1.0f / inv_b2 : 0.f;
    const float kmin = __expf(-13.0f);  // value of a kernel entry whose exponent hit the lower clamp
    float csum[TN], csq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { csum[b] = 0.f; csq[b] = 0.f; }
    const bool full_rows = (m0 + BM <= g.M);  // block-uniform: no per-row predicates; columns are per-lane
    auto transform = [&](float v, float kf) -> float {
        if (g.epi == EPI_CHORD) {
            v = 2.0f - 2.0f * v;  // src/mean_shift.py:154 / :168
        } else if (g.epi == EPI_MSKERNEL) {
            // src/mean_shift.py:65-68: dist = 2 - 2 s; K = exp(clamp(-dist / b^2 / 2, -13, 75))
            const float dist = 2.0f - 2.0f * v;
            float t = (-dist * rcp_b2) * 0.5f;
            t = fminf(fmaxf(t, -13.0f), 75.0f);
            v = __expf(t);  // v_exp_f32 path: relative error < 1e-6 on [-13, 75]
        } else if (g.epi == EPI_MSBWD) {
            // backward of K = exp(clamp((s-1)/b^2)): dL/ds = dL/dK * K / b^2, zero where clamped
            v = kf > kmin ? v * kf * rcp_b2 : 0.f;
        }
        return v;
    };
    const float *AUX = g.epi == EPI_MSBWD ? g.aux + (long long)z * g.sAux : nullptr;
    const float *RADD = (g.epi == EPI_MSBWD && g.row_add) ? g.row_add + (long long)z * g.M : nullptr;
    const float *YP = g.epi == EPI_BNRED ? g.aux : nullptr;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn0 + 32 * b + li;
        const bool cok = col < g.N;
        const float bias = (g.bias && cok && ks == 0) ? g.bias[(long long)z * g.bias_stride + col] : 0.f;
        float rs = 0.f, rt = 0.f, rmu = 0.f, ris = 0.f;
        if (g.epi == EPI_BNRED && cok) { rs = g.red_scale[col]; rt = g.red_shift[col]; rmu = g.red_mean[col]; ris = g.red_invstd[col]; }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int rbase = m0 + wm0 + 32 * a + 4 * lh;
            if (full_rows) {
                if (!cok) continue;
                float kf[16], ra[16];
                if (g.epi == EPI_MSBWD) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        kf[r] = AUX[(long long)(rbase + (r & 3) + 8 * (r >> 2)) * g.ldaux + col];
                        ra[r] = RADD ? RADD[rbase + (r & 3) + 8 * (r >> 2)] : 0.f;
                    }
                } else if (g.epi == EPI_BNRED) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) kf[r] = YP[(long long)(rbase + (r & 3) + 8 * (r >> 2)) * g.ldaux + col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    const float v = transform(acc[a][b][r] + bias + (g.epi == EPI_MSBWD ? ra[r] : 0.f),
                                              g.epi == EPI_MSBWD ? kf[r] : 0.f);
                    if (g.epi == EPI_BNRED) {
                        const float gm = fmaf(kf[r], rs, rt) > 0.f ? v : 0.f;
                        csum[b] += gm;
                        csq[b] += gm * ((kf[r] - rmu) * ris);
                    } else {
                        csum[b] += v;
                        csq[b] += v * v;
                    }
                    float *dst = C + (long long)row * g.ldc + col;
                    if (g.accumulate) {
                        if (g.splitk > 1) unsafeAtomicAdd(dst, v);
                        else *dst += v;
                    } else *dst = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (!(cok && row < g.M)) continue;
                    const float kf = g.epi == EPI_MSBWD ? AUX[(long long)row * g.ldaux + col]
                                                        : (g.epi == EPI_BNRED ? YP[(long long)row * g.ldaux + col] : 0.f);
                    const float v = transform(acc[a][b][r] + bias + (RADD ? RADD[row] : 0.f), kf);
                    if (g.epi == EPI_BNRED) {
                        const float gm = fmaf(kf, rs, rt) > 0.f ? v : 0.f;
                        csum[b] += gm;
                        csq[b] += gm * ((kf - rmu) * ris);
                    } else {
                        csum[b] += v;
                        csq[b] += v * v;
                    }
                    float *dst = C + (long long)row * g.ldc + col;
                    if (g.accumulate) {
                        if (g.splitk > 1) unsafeAtomicAdd(dst, v);
                        else *dst += v;
                    } else *dst = v;
                }
            }
        }
    }
    const bool tail = g.tail.acc != nullptr;
    if (g.stats || tail) {
        __syncthreads();  // LDS tiles are dead: reuse as [WAVES_M][2][BN]
        float *red = lds;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            float s = csum[b] + __shfl_xor(csum[b], 32, 64);
            float q = csq[b] + __shfl_xor(csq[b], 32, 64);
            if (lh == 0) {
                red[((wave / WAVES_N) * 2 + 0) * BN + wn0 + 32 * b + li] = s;
                red[((wave / WAVES_N) * 2 + 1) * BN + wn0 + 32 * b + li] = q;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * BN; t += NTH) {
            const int which = t / BN, c = t - which * BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < BM / WM; ++w) s += red[(w * 2 + which) * BN + c];
            if (n0 + c < g.N) {
                if (tail) bn_tail_add(g.tail, which, n0 + c, s);
                else g.stats[((long long)tile_m * 2 + which) * g.N + n0 + c] = s;
            }
        }
        if (tail) bn_tail_finish(g.tail, reinterpret_cast<int *>(lds));   // (its first barrier: `red` is read; lds[0] becomes the flag)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent form of the 128 x 128 kernel for the shared-MLP products with ONE z slice and a plain store: the forward
// C = relu(bn(A)) W^T + b with the column statistics (NT) and the dA product with the BatchNorm-backward partials (NN,
// EPI_BNRED).  A grid of the resident workgroups (two per CU) walks the output tiles; the first k-tile of the NEXT
// output tile is requested before the last k-tile's MFMAs of the current one and staged after the epilogue, so a
// workgroup goes from one tile's matrix instructions to the next one's with only the store issue in between.
// (One workgroup per tile leaves the first loads' round trip and the epilogue outside any overlap, and the two
// co-resident workgroups of a CU, started together and equally long, run their MFMA phases and their memory phases in
// step: the time of such a launch was the SUM of its MFMA time and its HBM time -- 0.44 of either peak.)
// All addressing is buffer addressing: rows / columns beyond the operands read as zeros and stores beyond C are dropped
// by the bounds check, the k-tile and the accumulator row are scalar offsets -- ragged extents (K = 196, N = 196) cost a
// compare per load in the last k-tile only, and the epilogue needs two address registers instead of sixty-four.
// PMAX (forward of a max-pooled last layer): per 32-row block and column, the largest and the smallest stored C and the
// row (0..31) of their first occurrence: cand[M / 32][4][N] = (max, argmax, min, argmin; indices as int bits), read by
// prifit_pool_from_candidates instead of C (as the streaming kernel does, csrc/gemm_stream.hip).
// DIAGNOSIS build (PRIFIT_BUILD_DEFS=-DPERS_STAMPS, tools/pers_stamps.py): wave 0 of eight workgroups stamps s_memtime at
// six phase boundaries of its first eight tiles.
#ifdef PERS_STAMPS
__device__ unsigned long long g_pers_stamps[8 * 8 * 8];   // [workgroup slot][tile][phase]
__device__ int g_pers_ktail = 1;                           // A/B inside one process: skip the empty k groups of the last k-tile
#define PERS_STAMP(i)                                                                                       \
    do {                                                                                                     \
        if (st_slot >= 0 && st_tile < 8) {                                                                   \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
            if (threadIdx.x == 0) g_pers_stamps[(st_slot * 8 + st_tile) * 8 + (i)] = t_;                    \
        }                                                                                                    \
    } while (0)
#else
#define PERS_STAMP(i) do { } while (0)
#endif

template <int LAY, bool FA, bool RED, bool PMAX = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_pers_kernel(const GemmArgs g)
{
#ifdef PERS_STAMPS
    const int st_slot = (blockIdx.x % 61 == 0 && blockIdx.x / 61 < 8) ? (int)blockIdx.x / 61 : -1;
    int st_tile = 0;
#endif
    constexpr int BM = 128, BN = 128, WM = 32, WN = 64, NTH = 512, TN = WN / 32;
    constexpr bool B_KC = (LAY == LAY_NT);
    constexpr int SZA = BM * (BK + PAD);
    constexpr int SZB = B_KC ? BN * (BK + PAD) : BK * (BN + PAD);
    __shared__ __attribute__((aligned(16))) float lds[2 * (SZA + SZB)];  // two stages
    // (an epilogue that transposes C through LDS for 128-bit row stores and forms the BatchNorm-backward partials in that
    // layout measured +3-4 % stand-alone and nothing inside the training step -- DESIGN 5c -- and is not kept)
    static_assert((BM / WM) * 2 * BN <= 2 * (SZA + SZB), "the statistics slab fits the stages");

    const int tilesN = (g.N + BN - 1) / BN;
    const int nwg = g.ntiles;
    auto tile_of = [&](int bid, int &tm, int &tn) {   // XCD-aware bijective remap (see gemm_kernel); bid & 7 is kept by the grid stride
        const int xcd = bid & 7, pos = bid >> 3, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
        tm = L / tilesN; tn = L - tm * tilesN;
    };
    int tix = blockIdx.x, tile_m, tile_n;
    tile_of(tix, tile_m, tile_n);
    const int ktiles = (g.K + BK - 1) / BK;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    // wave w holds row group w & 3 and column half w >> 2: a SIMD (waves w and w + 4) owns one wave of each column half, so
    // the matrix work a ragged last column tile leaves out (below) is left out evenly over the four SIMDs
    constexpr int WAVES_M = BM / WM;
    const int wmi = wave % WAVES_M;
    const int wm0 = wmi * WM, wn0 = (wave / WAVES_M) * WN;
#ifdef PERS_STAMPS
    const int ngk_last = g_pers_ktail ? (g.K - (ktiles - 1) * BK + 7) >> 3 : BK / 8;
#else
    const int ngk_last = (g.K - (ktiles - 1) * BK + 7) >> 3;   // 8-wide k groups of the last k-tile that hold data
#endif

    BufLoader<BM, true, FA, NTH> la;
    BufLoader<BN, B_KC, false, NTH> lb;
    la.init(g.A, g.lda, tile_m * BM, g.M, g.K);
    lb.init(g.B, g.ldb, tile_n * BN, g.N, g.K);
    la.load(0); lb.load(0);
    la.finish(g.a_scale, g.a_shift);
    la.store(lds); lb.store(lds + SZA);
    __syncthreads();

    // (C == NULL, the pooled forward whose backward does not read Y: an empty resource -- the bounds check drops every store)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.C ? (int)((long long)g.M * g.ldc * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(RED ? g.aux : nullptr), 0, RED ? (int)((long long)g.M * g.ldaux * 4) : 0, 0x00020000);
    const int ldc4 = (int)g.ldc * 4, ldy4 = RED ? (int)g.ldaux * 4 : 0;

    for (;;) {
        PERS_STAMP(0);
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const int ntix = tix + (int)gridDim.x;
        const bool has_next = ntix < nwg;
        int ntile_m = 0, ntile_n = 0;
        if (has_next) tile_of(ntix, ntile_m, ntile_n);
        f32x16 acc[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        const int nbv = min(TN, max(0, (g.N - n0 - wn0 + 31) >> 5));   // this wave's column blocks that hold columns of C
        for (int kt = 0; kt < ktiles; ++kt) {
            const int stage = kt & 1;
            const float *As = lds + stage * (SZA + SZB), *Bs = As + SZA;
            const bool more = kt + 1 < ktiles;
            if (more) {
                la.load((kt + 1) * BK); lb.load((kt + 1) * BK);
            } else if (has_next) {
                la.init(g.A, g.lda, ntile_m * BM, g.M, g.K);
                lb.init(g.B, g.ldb, ntile_n * BN, g.N, g.K);
                la.load(0); lb.load(0);
            }
            // (8-wide k groups beyond K hold zeros: K = 196 fills one of the last k-tile's four)
            mma_ktile<BM, true, BN, B_KC, TN>(As, Bs, wm0 + li, wn0 + li, lh, acc, more ? BK / 8 : ngk_last, nbv);
            if (more) {
                float *An = lds + (stage ^ 1) * (SZA + SZB);
                la.finish(g.a_scale, g.a_shift);
                la.store(An); lb.store(An + SZA);
            }
            __syncthreads();
#ifdef PERS_STAMPS
            if (kt == ktiles - 2) PERS_STAMP(1);   // all but the last k-tile
#endif
        }
        PERS_STAMP(2);

        // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const bool full_rows = m0 + BM <= g.M;   // block-uniform
        const int rbase = m0 + wm0 + 4 * lh;
        float csum[TN], csq[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            csum[b] = 0.f; csq[b] = 0.f;
            const int col = n0 + wn0 + 32 * b + li;
            const bool cok = col < g.N;
            const float bias = (g.bias && cok) ? g.bias[col] : 0.f;
            float rs = 0.f, rt = 0.f, rmu = 0.f, ris = 0.f;
            if (RED && cok) { rs = g.red_scale[col]; rt = g.red_shift[col]; rmu = g.red_mean[col]; ris = g.red_invstd[col]; }
            // lane part of the address (the first of this lane's rows, its column); not an existing column: fails the bounds check
            const int c_voff = cok ? (int)(((long long)rbase * g.ldc + col) * 4) : 0x7fffffff;
            const int y_voff = (RED && cok) ? (int)(((long long)rbase * g.ldaux + col) * 4) : 0x7fffffff;
            float kf[16];
            if (RED) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = ((r & 3) + 8 * (r >> 2)) * ldy4;
                    kf[r] = __builtin_bit_cast(float, full_rows ? __builtin_amdgcn_raw_buffer_load_b32(yrs, y_voff, ro, 0)
                                                                 : __builtin_amdgcn_raw_buffer_load_b32(yrs, cok ? y_voff + ro : y_voff, 0, 0));
                }
            }
            float vmax = -INFINITY, vmin = INFINITY;
            int imax = 0, imin = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const float v = acc[b][r] + bias;
                const float vs = (cok && (full_rows || row < g.M)) ? v : 0.f;
                if (PMAX) {  // rows ascend with r inside a lane: strict comparisons keep the first occurrence
                    const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool rok = full_rows || row < g.M;
                    if (rok && v > vmax) { vmax = v; imax = ri; }
                    if (rok && v < vmin) { vmin = v; imin = ri; }
                }
                if (RED) {
                    const float gm = fmaf(kf[r], rs, rt) > 0.f ? vs : 0.f;
                    csum[b] += gm;
                    csq[b] += gm * ((kf[r] - rmu) * ris);
                } else {
                    csum[b] += vs;
                    csq[b] += vs * vs;
                }
                {
                    const int ro = ((r & 3) + 8 * (r >> 2)) * ldc4;
                    if (full_rows) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, c_voff, ro, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, cok ? c_voff + ro : c_voff, 0, 0);
                }
            }
            if (PMAX) {
                // the other half of the block's rows lives in lane ^ 32; ties go to the lower row
                const float ov = __shfl_xor(vmax, 32, 64), on = __shfl_xor(vmin, 32, 64);
                const int oi = __shfl_xor(imax, 32, 64), oj = __shfl_xor(imin, 32, 64);
                if (ov > vmax || (ov == vmax && oi < imax)) { vmax = ov; imax = oi; }
                if (on < vmin || (on == vmin && oj < imin)) { vmin = on; imin = oj; }
                const int blk_row = m0 + wm0;
                if (lh == 0 && cok && blk_row < g.M) {
                    float *cd = g.cand + (long long)(blk_row >> 5) * 4 * g.N + col;
                    cd[0] = vmax; cd[g.N] = __int_as_float(imax); cd[2 * g.N] = vmin; cd[3 * g.N] = __int_as_float(imin);
                }
            }
        }
        PERS_STAMP(3);
        if (g.stats || g.tail.acc) {
            float *red = lds;   // both stages are dead: everybody passed the barrier that ended the k-loop
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const float s = csum[b] + __shfl_xor(csum[b], 32, 64);
                const float q = csq[b] + __shfl_xor(csq[b], 32, 64);
                if (lh == 0) {
                    red[(wmi * 2 + 0) * BN + wn0 + 32 * b + li] = s;
                    red[(wmi * 2 + 1) * BN + wn0 + 32 * b + li] = q;
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < 2 * BN; t += NTH) {
                const int which = t / BN, c = t - which * BN;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < BM / WM; ++w) s += red[(w * 2 + which) * BN + c];
                if (n0 + c < g.N) {
                    if (g.tail.acc) bn_tail_add(g.tail, which, n0 + c, s);
                    else g.stats[((long long)tile_m * 2 + which) * g.N + n0 + c] = s;
                }
            }
            if (has_next || g.tail.acc) __syncthreads();   // `red` is read before the next tile is staged over it
        }
        PERS_STAMP(4);
        if (!has_next) break;
        tix = ntix; tile_m = ntile_m; tile_n = ntile_n;
        la.finish(g.a_scale, g.a_shift);
        la.store(lds); lb.store(lds + SZA);
        __syncthreads();
        PERS_STAMP(5);
#ifdef PERS_STAMPS
        ++st_tile;
#endif
    }
    if (g.tail.acc) bn_tail_finish(g.tail, reinterpret_cast<int *>(lds));   // (after the last tile's barrier: lds is free)
}

// ---------------------------------------------------------------------------------------------------------------------
// Chord-distance matrix of ONE point set with itself (src/mean_shift.py:154 / :185: 2 - 2 X X^T for the bandwidth
// statistic and the final non-maximum suppression): C[z] = 2 - 2 A[z] A[z]^T, [n, n] per shape, n % 128 == 0, K % 32 == 0.
// The matrix is symmetric and so is its arithmetic -- C[i][j] and C[j][i] are the same products added in the same
// k order -- so only the tiles on and above the diagonal are computed (136 of 256 at n = 2048); an off-diagonal tile
// is also written transposed, through LDS, as whole coalesced rows.  Bit-identical to the full product.
#ifndef CHORD_XCD
#define CHORD_XCD 1     // (A/B builds: -DCHORD_XCD=0 is round 4's (tile, shape) launch order)
#endif
// MODE 1 (GRAM): the plain product A A^T instead (the pairwise matrix of src/dgcnn.py:13 for the second neighbour graph, K = 64).
// MODE 2 (MASK): what nms reads of the matrix once its owner pass is fused in (okey) is ONE BIT per element -- `dist[u][j] < b`
// in the neighbour pick, src/mean_shift.py:185-190 -- so the matrix is not written at all: mask[z][row][col / 32] bit col % 32
// = (2 - 2 a_row . a_col < thr[z]), 12.6 MB instead of 403 MB at 24 x 2048 x 2048; the same comparison of the same float.
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void chord_sym_kernel(
    const float *__restrict__ A, long long lda, long long sA, float *__restrict__ C, long long ldc, long long sC, int n, int K,
    int batch, unsigned long long *__restrict__ okey, const float *__restrict__ thr, unsigned *__restrict__ mask)
{
    constexpr bool GRAM = MODE == 1, MASK = MODE == 2;
    constexpr int BM = 128, BN = 128, WM = 32, WN = 64, NTH = 512, TN = WN / 32, WAVES_N = BN / WN;
    constexpr int SZ = BM * (BK + PAD);
    constexpr int TLD = BM + 4;   // transposed staging: [column of the tile][row], 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float lds[4 * SZ];   // two stages of (row panel, column panel); 73.7 KB >= 128 * TLD * 4
    static_assert(4 * SZ >= BN * TLD, "the transposed tile fits the staging buffers");
    // upper-triangular tile index -> (tile_m <= tile_n).  Round 5: all tiles of a shape on ONE XCD (xcd_shape_block; the
    // grid is linear) -- in (tile, shape) launch order every XCD's L2 fetched every shape's rows: 217 MB of reads for a 25 MB
    // input by the PMC counters (1.48 x the algorithmic bytes of the launch)
    const int T = n / BM;
    int t, zb;
#if CHORD_XCD
    xcd_shape_block((int)blockIdx.x, T * (T + 1) / 2, batch, zb, t);
#else
    t = (int)blockIdx.x % (T * (T + 1) / 2); zb = (int)blockIdx.x / (T * (T + 1) / 2);
#endif
    int tile_m = 0;
    while (t >= T - tile_m) { t -= T - tile_m; ++tile_m; }
    const int tile_n = tile_m + t;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float *Az = A + (long long)zb * sA;
    float *Cz = MASK ? nullptr : C + (long long)zb * sC;
    const int MW = n / 32;                                       // mask words per row
    unsigned *Mz = MASK ? mask + (long long)zb * n * MW : nullptr;
    const float th = MASK ? thr[zb] : 0.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    BufLoader<BM, true, false, NTH> la, lb;
    la.init(Az, lda, m0, n, K);
    lb.init(Az, lda, n0, n, K);
    const int ktiles = K / BK;
    la.load(0); lb.load(0);
    la.store(lds); lb.store(lds + SZ);
    __syncthreads();
    f32x16 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    for (int kt = 0; kt < ktiles; ++kt) {
        const int stage = kt & 1;
        const float *As = lds + stage * 2 * SZ, *Bs = As + SZ;
        const bool more = kt + 1 < ktiles;
        if (more) { la.load((kt + 1) * BK); lb.load((kt + 1) * BK); }
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            const float4 fa = read_frag<BM, true>(As, wm0 + li, gk, lh);
            float4 fb[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = read_frag<BN, true>(Bs, wn0 + 32 * b + li, gk, lh);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[b].x, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[b].y, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[b].z, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[b].w, acc[b], 0, 0, 0);
            }
        }
        if (more) {
            float *An = lds + (stage ^ 1) * 2 * SZ;
            la.store(An); lb.store(An + SZ);
        }
        __syncthreads();
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool mirror = tile_m != tile_n;   // block-uniform
    // okey (optional): nms's owner pass fused in (src/mean_shift.py:168-170: owner[j] = argmin_i dist[i][j], first minimum).
    // Every point keeps a 64-bit key (order-preserving image of the distance << 32 | candidate index) whose minimum over
    // the launch IS (smallest distance, lowest index among equals); this tile contributes, for the points of its column
    // block, the candidates of its row block (a column minimum, from the accumulator registers) and -- off the diagonal --
    // for the points of its row block the candidates of its column block (from the transposed copy in LDS).  The matrix is
    // bitwise symmetric, so a column's minimum over rows is that point's row minimum over columns.
    unsigned long long *okz = okey ? okey + (long long)zb * n : nullptr;
    auto fkey = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = wn0 + 32 * b + li;
        float cmin = INFINITY;
        int cidx = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = GRAM ? acc[b][r] : 2.0f - 2.0f * acc[b][r];   // src/mean_shift.py:154 / :168
            if (MASK) {
                // the 64 lanes hold two rows x 32 columns of this register: one ballot = one mask word of each
                const unsigned long long m = __ballot(v < th);
                if (li == 0) Mz[(long long)(m0 + row) * MW + (n0 + wn0 + 32 * b) / 32] = lh ? (unsigned)(m >> 32) : (unsigned)m;
            } else {
                Cz[(long long)(m0 + row) * ldc + n0 + col] = v;
            }
            if (mirror) lds[col * TLD + row] = v;
            if (v < cmin) { cmin = v; cidx = row; }     // rows ascend with r inside a lane: strict keeps the first
        }
        if (okz) {
            const float ov = __shfl_xor(cmin, 32, 64);
            const int oi = __shfl_xor(cidx, 32, 64);
            if (ov < cmin || (ov == cmin && oi < cidx)) { cmin = ov; cidx = oi; }
            if (lh == 0) atomicMin(okz + n0 + col, ((unsigned long long)fkey(cmin) << 32) | (unsigned)(m0 + cidx));
        }
    }
    if (mirror) {
        __syncthreads();
        if (MASK) {
            // thread (row of the transposed tile, word): 32 staged values -> one mask word  (BN * BM / 32 = NTH words)
            static_assert(BN * (BM / 32) == NTH, "one word per thread");
            const int trow = threadIdx.x / (BM / 32), w = threadIdx.x - trow * (BM / 32);
            unsigned bits = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(lds + trow * TLD + 32 * w + 4 * q);
                bits |= (unsigned)(v.x < th) << (4 * q) | (unsigned)(v.y < th) << (4 * q + 1) | (unsigned)(v.z < th) << (4 * q + 2) |
                        (unsigned)(v.w < th) << (4 * q + 3);
            }
            Mz[(long long)(n0 + trow) * MW + m0 / 32 + w] = bits;
        } else
        for (int i = threadIdx.x; i < BN * (BM / 4); i += NTH) {
            const int trow = i / (BM / 4), c4 = i - trow * (BM / 4);
            *reinterpret_cast<float4 *>(Cz + (long long)(n0 + trow) * ldc + m0 + 4 * c4) =
                *reinterpret_cast<const float4 *>(lds + trow * TLD + 4 * c4);
        }
        if (okz) {
            // lds[c][r] = dist[m0 + r][n0 + c]: thread (r, quarter q) scans the columns 32 q .. 32 q + 31 of row r
            const int r = threadIdx.x & (BM - 1), q = threadIdx.x >> 7;
            float rmin = INFINITY;
            int ridx = 0;
#pragma unroll 8
            for (int c = 32 * q; c < 32 * q + 32; ++c) {
                const float v = lds[c * TLD + r];
                if (v < rmin) { rmin = v; ridx = c; }
            }
            atomicMin(okz + m0 + r, ((unsigned long long)fkey(rmin) << 32) | (unsigned)(n0 + ridx));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-K form of the 128 x 128 kernel for the weight gradients dW[M, N] (+)= dY[P, M]^T relu(bn(A[P, N])) (TN: both
// operands stream from HBM over the reduction index -- 32 KB per workgroup and k-tile, 4.6 TB/s at the full matrix
// rate).  The general kernel keeps ONE k-tile in flight per workgroup and moved ~3 TB/s on these products whatever the
// split; this one keeps TWO (a second register set: it has nothing but the k-loop and the atomic epilogue, 100 VGPRs).
template <bool FB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_tnsk_kernel(const GemmArgs g)
{
    constexpr int BM = 128, BN = 128, WM = 32, WN = 64, NTH = 512, TN = WN / 32;
    constexpr int SZA = BK * (BM + PAD), SZB = BK * (BN + PAD);
    __shared__ __attribute__((aligned(16))) float lds[2 * (SZA + SZB)];  // two stages
    const int tilesN = (g.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tilesN, tile_n = blockIdx.x - tile_m * tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = blockIdx.z * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;   // block-uniform
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave % (BM / WM)) * WM, wn0 = (wave / (BM / WM)) * WN;   // (one wave of each column half per SIMD)
    // column blocks of this wave that hold elements of C: none below the last row (M = 196: the fourth row group of the
    // second tile), not those beyond the last column (N = 196)
    const int nbv = m0 + wm0 >= g.M ? 0 : min(TN, max(0, (g.N - n0 - wn0 + 31) >> 5));

    BufLoader<BM, false, false, NTH> la[2];
    BufLoader<BN, false, FB, NTH> lb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { la[i].init(g.A, g.lda, m0, g.M, g.K); lb[i].init(g.B, g.ldb, n0, g.N, g.K); }
    // tile kt0 -> LDS stage 0; tile kt0 + 1 -> register set 1 (in flight)
    la[0].load(kt0 * BK); lb[0].load(kt0 * BK);
    if (kt0 + 1 < kt1) { la[1].load((kt0 + 1) * BK); lb[1].load((kt0 + 1) * BK); }
    lb[0].finish(g.b_scale, g.b_shift);
    la[0].store(lds); lb[0].store(lds + SZA);
    __syncthreads();
    f32x16 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // iteration i (tile kt = kt0 + i): LDS stage i & 1 is multiplied, register set i & 1 receives tile kt + 2, register
    // set (i + 1) & 1 (tile kt + 1, requested one iteration ago) is staged into the other LDS stage
    auto step = [&](int kt, auto par) {
        constexpr int S = decltype(par)::value;
        const float *As = lds + S * (SZA + SZB), *Bs = As + SZA;
        if (kt + 2 < kt1) { la[S].load((kt + 2) * BK); lb[S].load((kt + 2) * BK); }
        __builtin_amdgcn_sched_barrier(0);
        mma_ktile<BM, false, BN, false, TN>(As, Bs, wm0 + li, wn0 + li, lh, acc, BK / 8, nbv);
        if (kt + 1 < kt1) {
            float *An = lds + (S ^ 1) * (SZA + SZB);
            lb[S ^ 1].finish(g.b_scale, g.b_shift);
            la[S ^ 1].store(An); lb[S ^ 1].store(An + SZA);
        }
        __syncthreads();
    };
    int kt = kt0;
    for (; kt + 1 < kt1; kt += 2) {
        step(kt, std::integral_constant<int, 0>());
        step(kt + 1, std::integral_constant<int, 1>());
    }
    if (kt < kt1) step(kt, std::integral_constant<int, 0>());
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int rbase = m0 + wm0 + 4 * lh;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn0 + 32 * b + li;
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (row >= g.M) continue;
            float *dst = g.C + (long long)row * g.ldc + col;
            if (g.splitk > 1) unsafeAtomicAdd(dst, acc[b][r]);
            else if (g.accumulate) *dst += acc[b][r];
            else *dst = acc[b][r];
        }
    }
}

// the split-K kernel's cases: TN, one batch item, no epilogue / statistics / bias / A prologue, 16-byte rows, spans < 2 GiB
static bool launch_tnsk(const GemmArgs &g, hipStream_t st)
{
    if (g.batch != 1 || g.a_scale || g.bias || g.stats || g.tail.acc || g.epi != EPI_NONE || g.a_rowsum || g.kswitch ||
        !(g.vecA && g.vecB) || (g.M & 3) || (g.N & 3) || (g.splitk > 1 && !g.accumulate) || g.splitk > 65535)
        return false;
    const long long lim = 0x7ff00000LL;
    if ((long long)g.K * g.lda * 4 >= lim || (long long)g.K * g.ldb * 4 >= lim) return false;
    const dim3 grid(((g.M + 127) / 128) * ((g.N + 127) / 128), 1, g.splitk), block(512);
    if (g.b_scale) hipLaunchKernelGGL((gemm_tnsk_kernel<true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_tnsk_kernel<false>), grid, block, 0, st, g);
    return true;
}

// the persistent kernel's cases: NT / NN, one z slice, plain store, (bias + statistics) or EPI_BNRED, 16-byte rows,
// 32-bit byte offsets everywhere, more tiles than resident workgroups
static bool launch_persistent(const GemmArgs &g_, int lay, hipStream_t st, float *want_cand = nullptr, bool dry = false)
{
    GemmArgs g = g_;
    if (lay == LAY_TN || g.batch != 1 || g.splitk != 1 || g.accumulate || g.a_rowsum || g.kswitch ||
        g.b_scale || !(g.epi == EPI_NONE || g.epi == EPI_BNRED) || !(g.vecA && g.vecB) || (g.K & 3) || (lay == LAY_NN && (g.N & 3)))
        return false;
    const long long lim = 0x7ff00000LL;
    if ((long long)g.M * g.lda * 4 >= lim || (long long)(lay == LAY_NT ? g.N : g.K) * g.ldb * 4 >= lim ||
        (long long)g.M * g.ldc * 4 >= lim || (g.epi == EPI_BNRED && (long long)g.M * g.ldaux * 4 >= lim))
        return false;
    g.ntiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int slots = 512;   // two 8-wave workgroups per CU (73 KB of LDS, <= 128 VGPRs)
    if (g.ntiles <= slots) return false;
    if (dry) return lay == LAY_NT && g.epi == EPI_NONE && g.a_scale;   // (prifit_gemm_pool_supported)
    const dim3 grid(slots), block(512);
    if (lay == LAY_NT) {
        if (g.epi != EPI_NONE) return false;
        if (want_cand) {
            g.cand = want_cand;
            if (g.a_scale) hipLaunchKernelGGL((gemm_pers_kernel<LAY_NT, true, false, true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_pers_kernel<LAY_NT, false, false, true>), grid, block, 0, st, g);
        } else if (g.a_scale) hipLaunchKernelGGL((gemm_pers_kernel<LAY_NT, true, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_pers_kernel<LAY_NT, false, false>), grid, block, 0, st, g);
    } else {
        if (want_cand) return false;
        if (g.a_scale) return false;
        if (g.epi == EPI_BNRED) hipLaunchKernelGGL((gemm_pers_kernel<LAY_NN, false, true>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_pers_kernel<LAY_NN, false, false>), grid, block, 0, st, g);
    }
    return true;
}

// Stream-K form of the dual-source NN product of the mean-shift backward (dX += gS^T Z + K^T gO; M = N_points, N = 128,
// K = 2 x N_points): the (tile, k-tile) work units of ALL tiles and batch items are laid out in one line and cut into
// equal contiguous ranges, one per PERSISTENT workgroup (grid = resident slots: 2 per CU).  The data-parallel launch has
// 16 x 24 = 384 output tiles x split-K 2 = 768 workgroups for 512 slots: one and a half rounds, i.e. a quarter of the
// machine idle on average (per-wave the kernel keeps its matrix pipe 88 % busy at full occupancy, PMC; 70 % overall);
// split-K 4 fills three rounds exactly but doubles the per-workgroup prologue / epilogue.  Here every workgroup runs
// 96 consecutive k-tiles (at most two segments of different tiles), each segment ends with float atomics into C.
// Same tile shape, loaders, fragment reads and k order inside a segment as gemm_kernel<128,128,32,64,LAY_NN,...,EX>.
__global__ __launch_bounds__(512, 4) void gemm_dual_sk_kernel(const GemmArgs g)
{
    constexpr int BM = 128, BN = 128, WN = 64, NTH = 512, TN = 2, WAVES_N = 2;
    constexpr int SZA = BM * (BK + PAD), SZB = BK * (BN + PAD);
    __shared__ __attribute__((aligned(16))) float lds[2 * (SZA + SZB)];
    const int tilesM = g.M / BM;
    const int ktiles = g.K / BK;
    const long long total = (long long)g.batch * tilesM * ktiles;
    int u = (int)(total * blockIdx.x / gridDim.x);
    const int uend = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WAVES_N) * 32, wn0 = (wave % WAVES_N) * WN;
    const int kswitch = g.kswitch ? g.kswitch : 0x7fffffff;

    while (u < uend) {
        const int t = u / ktiles;
        const int kt0 = u - t * ktiles, kt1 = min(ktiles, kt0 + (uend - u));
        const int z = t / tilesM, tile_m = t - z * tilesM;
        const int m0 = tile_m * BM;
        const float *A = g.A + (long long)z * g.sA;
        const float *B = g.B + (long long)z * g.sB;
        float *C = g.C + (long long)z * g.sC;

        f32x16 acc[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        ExactLoader<BM, true, false, NTH> la;
        ExactLoader<BN, false, false, NTH> lb;
        la.init(A, g.lda, m0, g.M, g.K);
        lb.init(B, g.ldb, 0, g.N, g.K);
        if (kt0 >= kswitch) { la.shift(g.dA2); lb.shift(g.dB2); }
        la.load(kt0 * BK, nullptr, nullptr);
        lb.load(kt0 * BK, nullptr, nullptr);
        la.store(lds);
        lb.store(lds + SZA);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int stage = (kt - kt0) & 1;
            const float *As = lds + stage * (SZA + SZB), *Bs = As + SZA;
            const bool more = kt + 1 < kt1;
            if (more) {
                if (kt + 1 == kswitch) { la.shift(g.dA2); lb.shift(g.dB2); }
                la.load((kt + 1) * BK, nullptr, nullptr);
                lb.load((kt + 1) * BK, nullptr, nullptr);
            }
            mma_ktile<BM, true, BN, false, TN>(As, Bs, wm0 + li, wn0 + li, lh, acc);
            if (more) {
                float *An = lds + (stage ^ 1) * (SZA + SZB);
                la.store(An);
                lb.store(An + SZA);
            }
            __syncthreads();
        }
        // segment epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); always atomics
        const int rbase = m0 + wm0 + 4 * lh;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = wn0 + 32 * b + li;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                unsafeAtomicAdd(C + (long long)(rbase + (r & 3) + 8 * (r >> 2)) * g.ldc + col, acc[b][r]);
        }
        u += kt1 - kt0;
    }
}

template <int BM, int BN, int WM, int WN, int LAY, bool VEC>
static void launch_aff(const GemmArgs &g, dim3 grid, hipStream_t st)
{
    // prologue combinations that occur: none, A only (forward / dA never has one), B only (dW)
    const dim3 block((BM / WM) * (BN / WN) * 64);
    // 16-byte rows, each operand (one batch item) below 2 GiB, no A row sums: the buffer loaders
    const long long spanA = (LAY == LAY_TN ? (long long)g.K * g.lda : (long long)g.M * g.lda) * 4;
    const long long spanB = (LAY == LAY_NT ? (long long)g.N * g.ldb : (long long)g.K * g.ldb) * 4;
    const bool exact = VEC && !g.a_rowsum && spanA < 0x7ff00000LL && spanB < 0x7ff00000LL;
    if (exact) {
        if (g.a_scale) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LAY, true, true, true, false, true>), grid, block, 0, st, g);
        else if (g.b_scale) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LAY, true, true, false, true, true>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LAY, true, true, false, false, true>), grid, block, 0, st, g);
        return;
    }
    if (g.a_scale) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LAY, VEC, VEC, true, false>), grid, block, 0, st, g);
    else if (g.b_scale) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LAY, VEC, VEC, false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LAY, VEC, VEC, false, false>), grid, block, 0, st, g);
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs &g, int lay, hipStream_t st)
{
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    dim3 grid(tilesM * tilesN, 1, g.batch * g.splitk);
    const bool vec = g.vecA && g.vecB;  // both or neither (mixed alignment takes the scalar path)
    if (g.a_scale && g.b_scale) return PRIFIT_EINVAL;
    switch (lay) {
        case LAY_NT:
            if (vec) launch_aff<BM, BN, WM, WN, LAY_NT, true>(g, grid, st);
            else launch_aff<BM, BN, WM, WN, LAY_NT, false>(g, grid, st);
            break;
        case LAY_NN:
            if (vec) launch_aff<BM, BN, WM, WN, LAY_NN, true>(g, grid, st);
            else launch_aff<BM, BN, WM, WN, LAY_NN, false>(g, grid, st);
            break;
        default:
            if (vec) launch_aff<BM, BN, WM, WN, LAY_TN, true>(g, grid, st);
            else launch_aff<BM, BN, WM, WN, LAY_TN, false>(g, grid, st);
            break;
    }
    return prifit_check_launch();
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// Few output tiles (the group-all / feature-propagation layers: M = B x 128 rows): 64 x 64 tiles put four times as
// many workgroups on the 256 CUs (M = 3072, N = 256: 48 -> 192).
static inline bool small_tiles(int M, int N, int batch, int splitk)
{
    const long long wgs = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch * splitk;
    // (fewer than ~two 128 x 128 tiles per CU: measured over the step's 28 SA3 / feature-propagation / head products,
    // tools/small_gemm_bench.py: 838 us with the threshold at 160 workgroups, 796 at 320, 786 at 520)
    // (split-K launches -- the tall weight-gradient reductions -- keep the 128 x 128 split-K kernel down to 160 workgroups)
    return wgs < (splitk > 1 ? 160 : 520) && M > 64 && N > 32;
}

static int dispatch(GemmArgs &g, int layout, void *stream)
{
    const int M = g.M, N = g.N, K = g.K;
    // contiguous extents: A is k-contiguous unless TN (then m-contiguous); B is k-contiguous for NT else n-contiguous
    const int extA = layout == LAY_TN ? M : K, extB = layout == LAY_NT ? K : N;
    g.vecA = aligned16(g.A) && (g.lda % 4 == 0) && (g.sA % 4 == 0) && (extA % 4 == 0);
    g.vecB = aligned16(g.B) && (g.ldb % 4 == 0) && (g.sB % 4 == 0) && (extB % 4 == 0);
    hipStream_t st = as_stream(stream);
    // N tile: 128 (2x2 waves of 64x64), 96 / 64 / 32 (4 waves stacked along M, each 32 x BN)
    if (small_tiles(M, N, g.batch, g.splitk)) return launch_cfg<64, 64, 32, 32>(g, layout, st);
    if (N <= 32) return launch_cfg<128, 32, 32, 32>(g, layout, st);
    if (N <= 64) return launch_cfg<128, 64, 32, 64>(g, layout, st);
    if (N <= 96) return launch_cfg<128, 96, 32, 96>(g, layout, st);
    if (launch_persistent(g, layout, st)) return prifit_check_launch();
    if (layout == LAY_TN && launch_tnsk(g, st)) return prifit_check_launch();
    return launch_cfg<128, 128, 32, 64>(g, layout, st);  // 8 waves of 32x64: more waves per SIMD hide the staging waits
}

extern "C" {

#ifdef PERS_STAMPS
int prifit_debug_pers_ktail(int on)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pers_ktail), &on, sizeof(int)) == hipSuccess ? 0 : -1;
}
int prifit_debug_pers_stamps(unsigned long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pers_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

int prifit_gemm_tile_m(int N)
{
    (void)N;
    return 128;
}

int prifit_gemm_stats_tile_m(int M, int N)
{
    return small_tiles(M, N, 1, 1) ? 64 : 128;
}

int prifit_gemm_f32(int layout, int M, int N, int K, const float *A, long long lda, long long strideA,
                    const float *B, long long ldb, long long strideB, float *C, long long ldc,
                    long long strideC, int batch, int splitk, const float *a_scale, const float *a_shift,
                    const float *b_scale, const float *b_shift, const float *bias, long long bias_batch_stride,
                    float *col_stats, int epilogue, const float *epi_batch_scalar, const float *epi_aux, long long ld_aux,
                    long long stride_aux, const float *epi_row_add, float *a_rowsum, int accumulate,
                    const prifit_bn_fwd *bn, void *stream)
{
    const bool has_tail = bn && bn->acc;
    if (bn_fwd_bad(bn) || (has_tail && (batch != 1 || splitk != 1 || epilogue != EPI_NONE))) return PRIFIT_EINVAL;
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || splitk <= 0 || layout < 0 || layout > 2 ||
        (epilogue != 100 && (epilogue < 0 || epilogue > 3)) || (epilogue >= EPI_MSKERNEL && !epi_batch_scalar) ||
        (epilogue == EPI_MSBWD && !epi_aux) || (splitk > 1 && !accumulate) ||
        ((a_scale == nullptr) != (a_shift == nullptr)) || ((b_scale == nullptr) != (b_shift == nullptr)) ||
        (col_stats && (batch != 1 || splitk != 1)) || (epilogue != EPI_NONE && epilogue != 100 && splitk != 1) || (long long)batch * splitk > 65535)
        return PRIFIT_EINVAL;
    GemmArgs g;
    g.tail = BnTail{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.batch = batch; g.splitk = splitk;
    g.a_scale = a_scale; g.a_shift = a_shift; g.b_scale = b_scale; g.b_shift = b_shift;
    g.bias = bias; g.bias_stride = bias_batch_stride; g.stats = col_stats; g.epi = epilogue; g.epi_batch_scalar = epi_batch_scalar;
    g.accumulate = accumulate; g.aux = epi_aux; g.ldaux = ld_aux; g.sAux = stride_aux;
    g.row_add = epi_row_add; g.a_rowsum = a_rowsum;
    g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr;
    g.dA2 = g.dB2 = 0; g.kswitch = 0;
    g.tail = bn_tail_fwd(bn, N);
    if (a_rowsum && (layout == LAY_TN || splitk != 1)) return PRIFIT_EINVAL;
    return dispatch(g, layout, stream);
}

int prifit_gemm_dual_nn_f32(int M, int N, int K1, int K2, const float *A1, const float *A2, long long lda,
                            long long strideA, const float *B1, const float *B2, long long ldb, long long strideB,
                            float *C, long long ldc, long long strideC, int batch, int splitk, int accumulate,
                            void *stream)
{
    if (!A1 || !A2 || !B1 || !B2 || !C || M <= 0 || N <= 0 || K1 <= 0 || K2 <= 0 || (K1 % BK) || batch <= 0 ||
        splitk <= 0 || (splitk > 1 && !accumulate) || (long long)batch * splitk > 65535 || lda < K1 || lda < K2 ||
        ldb < N || ldc < N)
        return PRIFIT_EINVAL;
    // the vector path needs both sources equally aligned
    if ((((uintptr_t)A1 ^ (uintptr_t)A2) | ((uintptr_t)B1 ^ (uintptr_t)B2)) & 15) return PRIFIT_EINVAL;
    GemmArgs g;
    g.tail = BnTail{};
    g.A = A1; g.B = B1; g.C = C; g.M = M; g.N = N; g.K = K1 + K2;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.batch = batch; g.splitk = splitk;
    g.a_scale = g.a_shift = g.b_scale = g.b_shift = nullptr;
    g.bias = nullptr; g.bias_stride = 0; g.stats = nullptr; g.epi = EPI_NONE; g.epi_batch_scalar = nullptr;
    g.accumulate = accumulate; g.aux = nullptr; g.ldaux = 0; g.sAux = 0; g.row_add = nullptr; g.a_rowsum = nullptr;
    g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr;
    // A is [M][k] (k contiguous): element (m, k >= K1) of the virtual operand lives at A2 + m*lda + (k - K1);
    // B is [k][N]: row k >= K1 lives at B2 + (k - K1)*ldb
    g.dA2 = (A2 - A1) - (long long)K1;
    g.dB2 = (B2 - B1) - (long long)K1 * ldb;
    g.kswitch = K1 / BK;
    // stream-K over persistent workgroups when the shape is whole tiles (the mean-shift backward: M = K1 = K2 = N_points)
    if (accumulate && N == 128 && M % 128 == 0 && K1 == K2 && aligned16(A1) && aligned16(B1) && (lda % 4 == 0) &&
        (ldb % 4 == 0) && (strideA % 4 == 0) && (strideB % 4 == 0) && (long long)M * lda * 4 < 0x7ff00000LL &&
        (long long)(K1 + K2) * ldb * 4 < 0x7ff00000LL) {
        static const int slots = [] {
            int dev = 0;
            hipDeviceProp_t p;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 512;
            return 2 * p.multiProcessorCount;   // two workgroups per CU fit (70.6 KB of LDS each)
        }();
        const long long total = (long long)batch * (M / 128) * ((K1 + K2) / BK);
        const int grid = (int)(total < slots ? total : slots);
        hipLaunchKernelGGL(gemm_dual_sk_kernel, dim3(grid), dim3(512), 0, as_stream(stream), g);
        return prifit_check_launch();
    }
    return dispatch(g, LAY_NN, stream);
}

int prifit_gemm_dgrad_bnred_f32(int M, int N, int K, const float *dY, long long lda, const float *W, long long ldb,
                                float *G, long long ldc, const float *Yprev, long long ldy, const float *scale,
                                const float *shift, const float *mean, const float *invstd, float *red_slab,
                                const prifit_bn_bwd *bn, void *stream)
{
    if (!dY || !W || !G || !Yprev || !scale || !shift || !mean || !invstd || (!red_slab && !(bn && bn->acc)) || bn_bwd_bad(bn) || M <= 0 || N <= 0 || K <= 0 ||
        lda < K || ldb < N || ldc < N || ldy < N)
        return PRIFIT_EINVAL;
    GemmArgs g;
    g.tail = BnTail{};
    g.A = dY; g.B = W; g.C = G; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = g.sB = g.sC = 0;
    g.batch = 1; g.splitk = 1;
    g.a_scale = g.a_shift = g.b_scale = g.b_shift = nullptr;
    g.bias = nullptr; g.bias_stride = 0; g.stats = red_slab; g.epi = EPI_BNRED; g.epi_batch_scalar = nullptr;
    g.accumulate = 0; g.aux = Yprev; g.ldaux = ldy; g.sAux = 0; g.row_add = nullptr; g.a_rowsum = nullptr;
    g.red_scale = scale; g.red_shift = shift; g.red_mean = mean; g.red_invstd = invstd;
    g.dA2 = g.dB2 = 0; g.kswitch = 0;
    g.tail = bn_tail_bwd(bn, N);
    return dispatch(g, LAY_NN, stream);
}

int prifit_chord_sym_f32(const float *A, long long lda, long long strideA, float *C, long long ldc, long long strideC,
                         int n, int K, int batch, unsigned long long *owner_key, void *stream)
{
    if (!A || !C || n <= 0 || (n % 128) || K <= 0 || (K % BK) || batch <= 0 || batch > 65535 || lda < K || ldc < n || (lda & 3) ||
        (ldc & 3) || (strideA & 3) || (strideC & 3) || ((uintptr_t)A & 15) || ((uintptr_t)C & 15) ||
        (long long)n * lda * 4 >= 0x7ff00000LL)
        return PRIFIT_EINVAL;
    const int T = n / 128;
    hipLaunchKernelGGL(chord_sym_kernel<0>, dim3((unsigned)(T * (T + 1) / 2 * batch)), dim3(512), 0, as_stream(stream), A, lda,
                       strideA, C, ldc, strideC, n, K, batch, owner_key, (const float *)nullptr, (unsigned *)nullptr);
    return prifit_check_launch();
}

int prifit_chord_sym_mask(const float *A, long long lda, long long strideA, const float *thr, uint32_t *mask, int n, int K, int batch,
                          unsigned long long *owner_key, void *stream)
{
    if (!A || !thr || !mask || n <= 0 || (n % 128) || K <= 0 || (K % BK) || batch <= 0 || batch > 65535 || lda < K || (lda & 3) ||
        (strideA & 3) || ((uintptr_t)A & 15) || (long long)n * lda * 4 >= 0x7ff00000LL)
        return PRIFIT_EINVAL;
    const int T = n / 128;
    hipLaunchKernelGGL(chord_sym_kernel<2>, dim3((unsigned)(T * (T + 1) / 2 * batch)), dim3(512), 0, as_stream(stream), A, lda,
                       strideA, (float *)nullptr, 0LL, 0LL, n, K, batch, owner_key, thr, mask);
    return prifit_check_launch();
}

int prifit_gram_sym_f32(const float *A, long long lda, long long strideA, float *C, long long ldc, long long strideC, int n, int K,
                        int batch, void *stream)
{
    if (!A || !C || n <= 0 || (n % 128) || K <= 0 || (K % BK) || batch <= 0 || batch > 65535 || lda < K || ldc < n || (lda & 3) ||
        (ldc & 3) || (strideA & 3) || (strideC & 3) || ((uintptr_t)A & 15) || ((uintptr_t)C & 15) ||
        (long long)n * lda * 4 >= 0x7ff00000LL)
        return PRIFIT_EINVAL;
    const int T = n / 128;
    hipLaunchKernelGGL(chord_sym_kernel<1>, dim3((unsigned)(T * (T + 1) / 2 * batch)), dim3(512), 0, as_stream(stream), A, lda,
                       strideA, C, ldc, strideC, n, K, batch, (unsigned long long *)nullptr, (const float *)nullptr, (unsigned *)nullptr);
    return prifit_check_launch();
}

static void pool_gemm_args(GemmArgs &g, int M, int N, int K, const float *A, long long lda, const float *W, long long ldb,
                           float *Y, long long ldc, const float *a_scale, const float *a_shift, const float *bias, float *stats)
{
    g.A = A; g.B = W; g.C = Y; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = g.sB = g.sC = 0; g.batch = 1; g.splitk = 1;
    g.a_scale = a_scale; g.a_shift = a_shift; g.b_scale = g.b_shift = nullptr;
    g.bias = bias; g.bias_stride = 0; g.stats = stats; g.epi = EPI_NONE; g.epi_batch_scalar = nullptr;
    g.aux = nullptr; g.row_add = nullptr; g.a_rowsum = nullptr; g.ldaux = 0; g.sAux = 0;
    g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr;
    g.dA2 = g.dB2 = 0; g.kswitch = 0; g.accumulate = 0;
    g.vecA = aligned16(A) && (lda % 4 == 0) && (K % 4 == 0);
    g.vecB = aligned16(W) && (ldb % 4 == 0) && (K % 4 == 0);
}

/* 1 when prifit_gemm_pool_f32 takes this shape (the persistent 128 x 128 kernel: more tiles than resident workgroups) */
int prifit_gemm_pool_supported(int M, int N, int K)
{
    if (M <= 0 || N <= 96 || K <= 0 || (M & 31)) return 0;
    GemmArgs g;
    g.tail = BnTail{};
    static float dummy[4] __attribute__((aligned(16)));
    pool_gemm_args(g, M, N, K, dummy, K, dummy, K, dummy, N, dummy, dummy, nullptr, nullptr);
    return launch_persistent(g, LAY_NT, nullptr, nullptr, true) ? 1 : 0;
}

int prifit_gemm_pool_f32(int M, int N, int K, const float *A, long long lda, const float *W, long long ldb, float *Y,
                         long long ldc, const float *a_scale, const float *a_shift, const float *bias, float *col_stats,
                         float *cand, const prifit_bn_fwd *bn, void *stream)
{
    if (bn_fwd_bad(bn)) return PRIFIT_EINVAL;
    if (!A || !W || ((a_scale == nullptr) != (a_shift == nullptr)) || !cand || !prifit_gemm_pool_supported(M, N, K) ||
        lda < K || ldb < K || ldc < N)
        return PRIFIT_EINVAL;
    GemmArgs g;
    g.tail = BnTail{};
    pool_gemm_args(g, M, N, K, A, lda, W, ldb, Y, ldc, a_scale, a_shift, bias, col_stats);
    g.tail = bn_tail_fwd(bn, N);
    if (!launch_persistent(g, LAY_NT, as_stream(stream), cand)) return PRIFIT_EINVAL;
    return prifit_check_launch();
}

}  // extern "C"
