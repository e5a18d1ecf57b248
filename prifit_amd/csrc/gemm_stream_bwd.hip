// Backward of one (conv1x1 + train-mode BatchNorm + ReLU) layer of a shared per-position MLP on the tall-and-skinny shapes
// (models/pointnet_util.py:195-199, :252-256 through autograd), BOTH products in one pass over the rows:
//     dY  = a (Y s + t > 0 ? G : 0) + (b Y + d)            middle layer   (bn_relu_bwd_apply's expression, term by term)
//         = b Y + d + (k == arg ? T : 0)                   max-pooled last layer (pool_bwd_apply's expression)
//     Gp  = dY W                  [P, Cin]   gradient w.r.t. relu(bn(Yp)) of the layer below, + the (m1, m2) column sums of
//                                            that layer's BatchNorm backward
//     dW  = dY^T relu(bn(Yp))     [Cout, Cin]
// The separate dA and dW kernels (gemm_stream_kernel NN, gemm_stream_tn_kernel) each stream dY's sources (G, Y) and Yp from
// HBM: 3.6 GB for SA1's widest middle layer where the tensors themselves are 2.0 GB.  Here a persistent workgroup stages
// every 64-row tile ONCE (dY formed while staging, Yp raw) and two groups of waves work on the same LDS tiles:
//   * A-waves (one per 32 columns of Cin) keep their Cout x 32 slice of W as MFMA fragments in registers and compute Gp
//     (k = Cout, fragments read along the rows of the dY tile), store it, and accumulate (m1, m2) from the Yp tile;
//   * W-waves accumulate their blocks of dW over ALL tiles of the workgroup in registers (k = the tile's rows: the A operand
//     is a column of the dY tile, the B operand a row of relu(s Yp + t), formed on read).
// One barrier per tile, the next tile in flight (global -> registers) during the MFMAs, two LDS stages.  Per-workgroup
// slabs for (m1, m2) and for dW, summed by the finalize kernel / a second small launch: deterministic, no atomics.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SBM = 64;   // rows per tile
#ifndef BWD_DEBUG
#define BWD_DEBUG 0   // timing-only builds: 1 no W-role MFMAs, 2 no A-role MFMAs, 3 no Gp stores, 4 no staging arithmetic
#endif

struct BwdArgs {
    long long P;
    const float *G, *Y;                       // [P, Cout] (G: NULL in POOL mode)
    const float *cs, *ct, *ca, *cb, *cd;      // per-channel [Cout]: mask scale / shift, a, b, d (cs, ct, ca: BN mode)
    const int32_t *pool_arg;                  // [P / pool_K, Cout]
    const float *pool_T;                      // [P / pool_K, Cout]
    int pool_K;
    const float *W;                           // [Cout, Cin] row-major, leading dimension ldw
    long long ldw;
    const float *Yp;                          // [P, Cin], leading dimension ldyp
    long long ldyp;
    const float *ps, *pt, *pmu, *pis;         // previous layer: scale, shift, mean, invstd [Cin]
    float *Gp;                                // [P, Cin], leading dimension ldgp
    long long ldgp;
    float *red_slab;                          // [grid][2][Cin]
    float *dw_part;                           // [grid][Cout][Cin]
    GatherSrc gs;                             // GATH: Yp is not stored, its rows are re-formed from (idx, U, Vc) -- common.h
    BnTail tail;                              // the (m1, m2) sums finalized by this launch instead of written to red_slab (common.h)
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Roles per (Cout, Cin): NA A-waves (one per 32 columns of Cin; Cout MFMAs per tile each) and W-waves that own a contiguous
// range of the (Cout / 32) x (Cin / 32) blocks of dW (32 MFMAs per block and tile), chosen so that the four SIMDs of a CU
// (wave w runs on SIMD w % 4) carry the same number of MFMAs per tile: (128, 96): A A A | W x 6, then W x 2, W x 2, W x 2
// beside the A-waves -- 192 per SIMD; (96, 64): A A | W x 3, W x 3; (128, 64): A A | W x 4, W x 4; (64, 64): A A | W x 2,
// W x 2; (128, 128): A A A A | W x 4 each.
template <int COUT, int CIN> struct Roles;
template <> struct Roles<128, 96> { static constexpr int NW = 7; static constexpr int wb0[7] = {0, 0, 0, 0, 6, 8, 10}, wcnt[7] = {0, 0, 0, 6, 2, 2, 2}; };
template <> struct Roles<96, 64> { static constexpr int NW = 4; static constexpr int wb0[4] = {0, 0, 0, 3}, wcnt[4] = {0, 0, 3, 3}; };
template <> struct Roles<128, 64> { static constexpr int NW = 4; static constexpr int wb0[4] = {0, 0, 0, 4}, wcnt[4] = {0, 0, 4, 4}; };
template <> struct Roles<64, 64> { static constexpr int NW = 4; static constexpr int wb0[4] = {0, 0, 0, 2}, wcnt[4] = {0, 0, 2, 2}; };
template <> struct Roles<128, 128> { static constexpr int NW = 8; static constexpr int wb0[8] = {0, 0, 0, 0, 0, 4, 8, 12}, wcnt[8] = {0, 0, 0, 0, 4, 4, 4, 4}; };

template <int V> struct IC { static constexpr int value = V; };

template <int COUT, int CIN, bool POOL, bool GATH = false>
__global__ __launch_bounds__((64 * Roles<COUT, CIN>::NW), 1) void gemm_stream_bwd_kernel(const BwdArgs g)
{
    using R = Roles<COUT, CIN>;
    static_assert(!GATH || (!POOL && (64 * Roles<COUT, CIN>::NW) % (CIN / 4) == 0), "GATH: one channel group per thread");
    constexpr int NA = CIN / 32, NTH = 64 * R::NW;
    constexpr int LDY = COUT + 4, LDP = CIN + 4;     // padded rows: conflict-free ds_read_b128 fragments / b32 columns
    constexpr int KG = COUT / 8;
    constexpr int Y4 = COUT / 4, P4 = CIN / 4;       // float4 per row
    constexpr int NY4 = SBM * Y4, NP4 = SBM * P4;
    constexpr int NVY = (NY4 + NTH - 1) / NTH, NVP = (NP4 + NTH - 1) / NTH;
    static_assert(!POOL || NTH % Y4 == 0, "POOL: one channel group per thread");
    __shared__ __attribute__((aligned(16))) float s_dy[2][SBM * LDY];
    __shared__ __attribute__((aligned(16))) float s_p[2][SBM * LDP];
    __shared__ __attribute__((aligned(16))) float s_co[5][COUT];   // s, t, a, b, d
    __shared__ int s_tail;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const bool a_wave = wave < NA;
    const int wa = a_wave ? wave : 0;

    for (int t = threadIdx.x; t < COUT; t += NTH) {
        s_co[0][t] = POOL ? 0.f : g.cs[t]; s_co[1][t] = POOL ? 0.f : g.ct[t]; s_co[2][t] = POOL ? 0.f : g.ca[t];
        s_co[3][t] = g.cb[t]; s_co[4][t] = g.cd[t];
    }

    const int tiles = (int)((g.P + SBM - 1) / SBM);
    float4 sty[NVY], stg[POOL ? 1 : NVY], stp[NVP];
    int4 st_arg = make_int4(0, 0, 0, 0);
    float4 st_T = make_float4(0.f, 0.f, 0.f, 0.f);
    int yoff[NVY], poff[NVP];
#pragma unroll
    for (int p = 0; p < NVY; ++p) {
        const int id = threadIdx.x + NTH * p;
        yoff[p] = id < NY4 ? ((id / Y4) * COUT + 4 * (id % Y4)) * 4 : 0x7fffffff;   // (beyond the tile: fails the bounds check)
    }
#pragma unroll
    for (int p = 0; p < NVP; ++p) {
        const int id = threadIdx.x + NTH * p;
        poff[p] = id < NP4 ? ((id / P4) * (int)g.ldyp + 4 * (id % P4)) * 4 : 0x7fffffff;
    }
    auto tile_rsrc = [&](const float *base, long long ld, int m0, int width) {
        const long long left = g.P - m0;
        const int rows = left < SBM ? (int)left : SBM;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + (long long)m0 * ld), 0,
                                                 ((rows - 1) * (int)ld + width) * 4, 0x00020000);
    };
    // GATH: point indices of the thread's Yp rows of the tile loaded next (requested one tile ahead), its centre term
    int nid[GATH ? NVP : 1];
    float4 gvc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int g_c4 = threadIdx.x % P4;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(GATH ? g.gs.U : nullptr), 0,
                                                                         GATH ? g.gs.ubytes : 0, 0x00020000);
    auto load_idx = [&](int tile) {
        if (GATH) {
            const int m0 = tile * SBM;
#pragma unroll
            for (int p = 0; p < NVP; ++p) {
                const int id = threadIdx.x + NTH * p;
                nid[p] = g.gs.idx[m0 + (id < NP4 ? id / P4 : 0)];   // (P % SBM == 0)
            }
        }
    };
    auto load_tile = [&](int tile) {
        const int m0 = tile * SBM;
        if (POOL) {
            const long long po = (long long)(m0 / g.pool_K) * COUT + 4 * (threadIdx.x % Y4);
            st_arg = *reinterpret_cast<const int4 *>(g.pool_arg + po);
            st_T = ld4(g.pool_T + po);
        }
        const __amdgpu_buffer_rsrc_t ry = tile_rsrc(g.Y, COUT, m0, COUT);
#pragma unroll
        for (int p = 0; p < NVY; ++p) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, yoff[p], 0, 0));
            sty[p] = make_float4(v.x, v.y, v.z, v.w);
        }
        if (!POOL) {
            const __amdgpu_buffer_rsrc_t rg = tile_rsrc(g.G, COUT, m0, COUT);
#pragma unroll
            for (int p = 0; p < NVY; ++p) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, yoff[p], 0, 0));
                stg[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
        if (GATH) {
            const int grp = m0 / g.gs.Kg;
            const int pbase = (grp / g.gs.S) * g.gs.N;
            gvc = ld4(g.gs.Vc + (long long)grp * CIN + 4 * g_c4);
#pragma unroll
            for (int p = 0; p < NVP; ++p) {
                const int id = threadIdx.x + NTH * p;
                const int n = nid[p];
                const int voff = id < NP4 ? ((pbase + ((n >= 0 && n < g.gs.N) ? n : 0)) * CIN + 4 * g_c4) * 4 : 0x7fffffff;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, voff, 0, 0));
                stp[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rp = tile_rsrc(g.Yp, g.ldyp, m0, CIN);
#pragma unroll
            for (int p = 0; p < NVP; ++p) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, poff[p], 0, 0));
                stp[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto store_tile = [&](int tile, int buf) {
        const int m0 = tile * SBM;
        const bool full = (long long)m0 + SBM <= g.P;   // block-uniform: only the tail tile pays for the row checks
#pragma unroll
        for (int p = 0; p < NVY; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (NVY * NTH != NY4 && id >= NY4) continue;
            const int row = id / Y4, c4 = id - row * Y4;
            float4 x = sty[p];
            const float4 cb4 = *reinterpret_cast<const float4 *>(&s_co[3][4 * c4]);
            const float4 cd4 = *reinterpret_cast<const float4 *>(&s_co[4][4 * c4]);
            if (POOL) {
                const int kr = m0 % g.pool_K + row;   // sample index of this row in its pooling group (a whole number of tiles)
                x.x = fmaf(cb4.x, x.x, cd4.x) + (st_arg.x == kr ? st_T.x : 0.f);
                x.y = fmaf(cb4.y, x.y, cd4.y) + (st_arg.y == kr ? st_T.y : 0.f);
                x.z = fmaf(cb4.z, x.z, cd4.z) + (st_arg.z == kr ? st_T.z : 0.f);
                x.w = fmaf(cb4.w, x.w, cd4.w) + (st_arg.w == kr ? st_T.w : 0.f);
            } else {
                const float4 cs = *reinterpret_cast<const float4 *>(&s_co[0][4 * c4]);
                const float4 ct = *reinterpret_cast<const float4 *>(&s_co[1][4 * c4]);
                const float4 ca = *reinterpret_cast<const float4 *>(&s_co[2][4 * c4]);
                const float4 gg = stg[p];
                x.x = fmaf(ca.x, fmaf(x.x, cs.x, ct.x) > 0.f ? gg.x : 0.f, fmaf(cb4.x, x.x, cd4.x));
                x.y = fmaf(ca.y, fmaf(x.y, cs.y, ct.y) > 0.f ? gg.y : 0.f, fmaf(cb4.y, x.y, cd4.y));
                x.z = fmaf(ca.z, fmaf(x.z, cs.z, ct.z) > 0.f ? gg.z : 0.f, fmaf(cb4.z, x.z, cd4.z));
                x.w = fmaf(ca.w, fmaf(x.w, cs.w, ct.w) > 0.f ? gg.w : 0.f, fmaf(cb4.w, x.w, cd4.w));
            }
            if (!full && (long long)m0 + row >= g.P) x = make_float4(0.f, 0.f, 0.f, 0.f);   // (the loads returned zeros; d is not zero)
            *reinterpret_cast<float4 *>(&s_dy[buf][row * LDY + 4 * c4]) = x;
        }
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int id = threadIdx.x + NTH * p;
            if (NVP * NTH != NP4 && id >= NP4) continue;
            const int row = id / P4, c4 = id - row * P4;
            float4 x = stp[p];
            if (GATH) { x.x -= gvc.x; x.y -= gvc.y; x.z -= gvc.z; x.w -= gvc.w; }   // y = U_j - Vc_g
            *reinterpret_cast<float4 *>(&s_p[buf][row * LDP + 4 * c4]) = x;
        }
    };
    // (GATH) the index list of the tile after `t`, requested behind t's loads: a whole iteration ahead of its use
    auto idx_after = [&](int t) {
        if (GATH) {
            const int n2 = t + (int)gridDim.x;
            load_idx(n2 < tiles ? n2 : t);
        }
    };

    int tile = blockIdx.x;
    if (tile >= tiles) tile = tiles - 1;   // (the launcher never starts more workgroups than tiles)
    load_idx(tile);
    load_tile(tile);
    idx_after(tile);
    __syncthreads();                        // s_co visible before the first staging
    store_tile(tile, 0);
    __syncthreads();
    // The two roles run the same loop skeleton (prefetch, MFMAs, stage the next tile, ONE barrier per tile) in separate
    // branches: the branch is wave-uniform, both sides execute the same barriers, and the register allocation is the
    // maximum of the two roles instead of their sum (W fragments 64 + dW accumulators 64 would not fit 256 VGPRs).
    if (a_wave) {
        // this wave's Cout x 32 slice of W as fragments: lane (li, lh) of k-group q holds W[8q + 4lh + 0..3][32 wa + li]
        const int col = 32 * wa + li;
        float4 bf[KG];
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            const int k0 = 8 * q + 4 * lh;
            bf[q] = make_float4(g.W[(long long)k0 * g.ldw + col], g.W[(long long)(k0 + 1) * g.ldw + col],
                                g.W[(long long)(k0 + 2) * g.ldw + col], g.W[(long long)(k0 + 3) * g.ldw + col]);
        }
        float r_s = g.ps[col], r_t = g.pt[col], r_mu = g.pmu[col], r_is = g.pis[col];
        // The fragments must LIVE in registers: as plain loads the compiler treats them as re-loadable (the stores to Gp may
        // alias W for all it knows) and fetched all 64 of them from global memory again in every tile, one s_waitcnt per
        // pair of MFMAs.  An empty asm makes each value opaque.
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            asm volatile("" : "+v"(bf[q].x), "+v"(bf[q].y), "+v"(bf[q].z), "+v"(bf[q].w));
        }
        asm volatile("" : "+v"(r_s), "+v"(r_t), "+v"(r_mu), "+v"(r_is));
        float m1 = 0.f, m2 = 0.f;
        const int ldgp4 = (int)g.ldgp * 4;
        const int c_voff = ((4 * lh) * (int)g.ldgp + col) * 4;
        for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
            const int cur = it & 1;
            const int next = tile + gridDim.x;
            const int ntile = next < tiles ? next : tile;
            load_tile(ntile);
            idx_after(ntile);
            __builtin_amdgcn_sched_barrier(0);   // the prefetch is issued HERE, ahead of the MFMAs
            f32x16 acc[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
            const float *ap = &s_dy[cur][li * LDY + 4 * lh];
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                float4 fa[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = *reinterpret_cast<const float4 *>(ap + a * 32 * LDY + 8 * q);
#if BWD_DEBUG != 2
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, bf[q].x, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, bf[q].y, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, bf[q].z, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, bf[q].w, acc[a], 0, 0, 0);
                }
#else
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a][q & 15] += fa[a].x * bf[q].x + fa[a].w * bf[q].w;
#endif
            }
            // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Rows beyond P
            // hold exact zeros (their dY rows are zero) and their stores are dropped by the bounds check.
            const __amdgpu_buffer_rsrc_t crs = tile_rsrc(g.Gp, g.ldgp, tile * SBM, CIN);
            const float *yp = &s_p[cur][(4 * lh) * LDP + col];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * a + (r & 3) + 8 * (r >> 2);
                    const float v = acc[a][r];
#if BWD_DEBUG != 3
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, c_voff, rl * ldgp4, 0);
#endif
                    const float y = yp[rl * LDP];
                    const float gm = fmaf(y, r_s, r_t) > 0.f ? v : 0.f;
                    m1 += gm;
                    m2 += gm * ((y - r_mu) * r_is);
                }
            // (nothing of the staging -- its arithmetic needs the prefetched tile -- may be scheduled up into the MFMAs: the
            // compiler did, and waited for the loads after the first eight matrix instructions)
            __builtin_amdgcn_sched_barrier(0);
            store_tile(ntile, cur ^ 1);
            __syncthreads();
        }
        m1 += __shfl_xor(m1, 32, 64);
        m2 += __shfl_xor(m2, 32, 64);
        if (lh == 0) {
            if (g.tail.acc) {
                bn_tail_add(g.tail, 0, col, m1);
                bn_tail_add(g.tail, 1, col, m2);
            } else {
                g.red_slab[((long long)blockIdx.x * 2 + 0) * CIN + col] = m1;
                g.red_slab[((long long)blockIdx.x * 2 + 1) * CIN + col] = m2;
            }
        }
    } else {
        // dW accumulators of this wave's blocks for the life of the workgroup; block id = (cout block) * NA + (cin block)
        auto w_role = [&](auto wb0c, auto wcntc) {
            constexpr int WB0 = decltype(wb0c)::value, WCNT = decltype(wcntc)::value;
            f32x16 accw[WCNT];
            float w_s[WCNT], w_t[WCNT];
#pragma unroll
            for (int i = 0; i < WCNT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
                w_s[i] = g.ps[32 * ((WB0 + i) % NA) + li];
                w_t[i] = g.pt[32 * ((WB0 + i) % NA) + li];
                asm volatile("" : "+v"(w_s[i]), "+v"(w_t[i]));   // (kept in registers, not re-loaded per tile: see the A role)
            }
            for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
                const int cur = it & 1;
                const int next = tile + gridDim.x;
                const int ntile = next < tiles ? next : tile;
                load_tile(ntile);
                idx_after(ntile);
                __builtin_amdgcn_sched_barrier(0);
                const float *dcol = &s_dy[cur][lh * LDY + li];
                const float *prow = &s_p[cur][lh * LDP + li];
                // operands of the NEXT group of k-steps are requested before the MFMAs of the current one: a W-wave may be
                // alone on its SIMD (timing-only builds: this loop, not the A role, set the pace -- 130 cycles per MFMA with
                // every operand read right in front of its use)
                constexpr int GS = WCNT >= 4 ? 2 : 4, NG = SBM / 2 / GS;
                float av[2][GS][WCNT], pv[2][GS][WCNT];
                auto fetch = [&](int buf, int grp) {
#pragma unroll
                    for (int u = 0; u < GS; ++u)
#pragma unroll
                        for (int i = 0; i < WCNT; ++i) {
                            av[buf][u][i] = dcol[2 * (grp * GS + u) * LDY + 32 * ((WB0 + i) / NA)];
                            pv[buf][u][i] = prow[2 * (grp * GS + u) * LDP + 32 * ((WB0 + i) % NA)];
                        }
                };
                fetch(0, 0);
#pragma unroll
                for (int grp = 0; grp < NG; ++grp) {
                    if (grp + 1 < NG) fetch((grp + 1) & 1, grp + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
#pragma unroll
                        for (int i = 0; i < WCNT; ++i) {
                            const float bv = fmaxf(fmaf(pv[grp & 1][u][i], w_s[i], w_t[i]), 0.f);
#if BWD_DEBUG != 1
                            accw[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[grp & 1][u][i], bv, accw[i], 0, 0, 0);
#else
                            accw[i][0] += av[grp & 1][u][i] * bv;
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);   // (as in the A role: the staging stays behind the MFMAs)
                store_tile(ntile, cur ^ 1);
                __syncthreads();
            }
            float *dst = g.dw_part + (long long)blockIdx.x * COUT * CIN;
#pragma unroll
            for (int i = 0; i < WCNT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * ((WB0 + i) / NA) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    dst[row * CIN + 32 * ((WB0 + i) % NA) + li] = accw[i][r];
                }
        };
        // (one instantiation per W-wave of the role table: block indices are compile-time, the branch is wave-uniform)
        if constexpr (R::NW > NA + 0) { if (wave == NA + 0) w_role(IC<R::wb0[NA + 0]>{}, IC<R::wcnt[NA + 0]>{}); }
        if constexpr (R::NW > NA + 1) { if (wave == NA + 1) w_role(IC<R::wb0[NA + 1]>{}, IC<R::wcnt[NA + 1]>{}); }
        if constexpr (R::NW > NA + 2) { if (wave == NA + 2) w_role(IC<R::wb0[NA + 2]>{}, IC<R::wcnt[NA + 2]>{}); }
        if constexpr (R::NW > NA + 3) { if (wave == NA + 3) w_role(IC<R::wb0[NA + 3]>{}, IC<R::wcnt[NA + 3]>{}); }
    }
    if (g.tail.acc) bn_tail_finish(g.tail, &s_tail);      // every wave of both roles arrives here
}

// dW[c] = sum over workgroups of their partial slabs, in a fixed order: 64 outputs per workgroup, four threads per output
// take every fourth slab (four independent loads in flight each), combined through LDS.  (One thread per output walking all
// 256 slabs was a 61 us launch of 24 workgroups: a serial chain of dependent adds on L2 latency.)
__global__ __launch_bounds__(256) void stream_bwd_dw_reduce_kernel(const float *__restrict__ part, int nwg, int n, int cin,
                                                                  long long lddw, float *__restrict__ dW)
{
    __shared__ float s_p[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int w = q;
        for (; w + 12 < nwg; w += 16) {
            a0 += part[(long long)w * n + i]; a1 += part[(long long)(w + 4) * n + i];
            a2 += part[(long long)(w + 8) * n + i]; a3 += part[(long long)(w + 12) * n + i];
        }
        for (; w < nwg; w += 4) a0 += part[(long long)w * n + i];
    }
    s_p[q][o] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && i < n) dW[(long long)(i / cin) * lddw + (i % cin)] = (s_p[0][o] + s_p[1][o]) + (s_p[2][o] + s_p[3][o]);
}

int bwd_grid(long long P, int Cout, int Cin)
{
    const long long tiles = (P + SBM - 1) / SBM;
    // LDS per workgroup: two stages of both tiles; two workgroups per CU where they fit in 160 KB
    const long long lds = 2LL * SBM * (Cout + 4 + Cin + 4) * 4 + 5 * Cout * 4 + 4 * Cin * 4;
    const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
    const long long want = 256LL * per_cu;
    return (int)(tiles < want ? tiles : want);
}

bool bwd_shape_ok(int Cout, int Cin)
{
    return (Cout == 128 && (Cin == 128 || Cin == 96 || Cin == 64)) || (Cout == 96 && Cin == 64) || (Cout == 64 && Cin == 64);
}

template <int COUT, int CIN>
void bwd_launch(const BwdArgs &g, bool pool, int grid, hipStream_t st)
{
    constexpr int NTH = 64 * Roles<COUT, CIN>::NW;
    if (g.gs.idx) {
        if constexpr (CIN == 64 && NTH % (CIN / 4) == 0) hipLaunchKernelGGL((gemm_stream_bwd_kernel<COUT, CIN, false, true>), dim3(grid), dim3(NTH), 0, st, g);
        return;
    }
    if (pool) {
        if constexpr (NTH % (COUT / 4) == 0) hipLaunchKernelGGL((gemm_stream_bwd_kernel<COUT, CIN, true>), dim3(grid), dim3(NTH), 0, st, g);
    } else {
        hipLaunchKernelGGL((gemm_stream_bwd_kernel<COUT, CIN, false>), dim3(grid), dim3(NTH), 0, st, g);
    }
}

}  // namespace

extern "C" {

int prifit_gemm_stream_bwd_supported(long long P, int Cout, int Cin, int pool_K)
{
    if (P < SBM || P > 0x7fffffffLL * 32 || !bwd_shape_ok(Cout, Cin)) return 0;
    if (pool_K > 0) {
        if (pool_K % SBM != 0 || P % pool_K != 0 || Cout == 96) return 0;   // (Cout = 96: a thread's rows would not share their channels)
    }
    return 1;
}

int prifit_gemm_stream_bwd_slabs(long long P, int Cout, int Cin) { return bwd_shape_ok(Cout, Cin) ? bwd_grid(P, Cout, Cin) : 0; }

long long prifit_gemm_stream_bwd_workspace(long long P, int Cout, int Cin)
{
    return bwd_shape_ok(Cout, Cin) ? (long long)bwd_grid(P, Cout, Cin) * Cout * Cin : 0;
}

static int stream_bwd_impl(long long P, int Cout, int Cin, const float *G, const float *Y, const float *scale,
                           const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                           const int32_t *pool_arg, const float *pool_T, int pool_K, const float *W, long long ldw,
                           const float *Yp, long long ldyp, const float *p_scale, const float *p_shift,
                           const float *p_mean, const float *p_invstd, float *Gp, long long ldgp, float *red_slab,
                           float *dW, long long lddw, float *workspace, const GatherSrc *gs, const prifit_bn_bwd *bn, void *stream)
{
    const bool pool = pool_arg != nullptr;
    if (bn_bwd_bad(bn)) return PRIFIT_EINVAL;
    if (gs) {   // Yp re-formed from (idx, U, Vc): middle layers on a 64-wide first layer only
        if (pool || Cin != 64 || !gs->idx || !gs->U || !gs->Vc || gs->N <= 0 || gs->S <= 0 || gs->Kg <= 0 || (gs->Kg % SBM) ||
            gs->C != Cin || P % ((long long)gs->S * gs->Kg) != 0 || (((uintptr_t)gs->U | (uintptr_t)gs->Vc) & 15) ||
            (P / ((long long)gs->S * gs->Kg)) * gs->N * Cin * 4 >= 0x7ff00000LL)
            return PRIFIT_EINVAL;
        Yp = gs->U;   // (only checked for presence and alignment below)
        ldyp = Cin;
    }
    if (!Y || !coef_b || !coef_d || !W || !Yp || !p_scale || !p_shift || !p_mean || !p_invstd || !Gp || (!red_slab && !(bn && bn->acc)) || !dW ||
        !workspace || !prifit_gemm_stream_bwd_supported(P, Cout, Cin, pool ? pool_K : 0) || ldw < Cin || ldyp < Cin || ldgp < Cin ||
        lddw < Cin || (ldyp & 3) || (pool ? (!pool_T) : (!G || !scale || !shift || !coef_a)) ||
        (((uintptr_t)Y | (uintptr_t)G | (uintptr_t)Yp | (uintptr_t)pool_arg | (uintptr_t)pool_T) & 15) ||
        (long long)SBM * ldyp * 4 >= 0x7ff00000LL || (long long)SBM * ldgp * 4 >= 0x7ff00000LL)
        return PRIFIT_EINVAL;
    BwdArgs g;
    g.P = P; g.G = G; g.Y = Y; g.cs = scale; g.ct = shift; g.ca = coef_a; g.cb = coef_b; g.cd = coef_d;
    g.pool_arg = pool_arg; g.pool_T = pool_T; g.pool_K = pool ? pool_K : 1;
    g.W = W; g.ldw = ldw; g.Yp = Yp; g.ldyp = ldyp; g.ps = p_scale; g.pt = p_shift; g.pmu = p_mean; g.pis = p_invstd;
    g.Gp = Gp; g.ldgp = ldgp; g.red_slab = red_slab; g.dw_part = workspace;
    g.tail = bn_tail_bwd(bn, Cin);
    g.gs.idx = nullptr;
    if (gs) {
        g.gs = *gs;
        g.gs.ubytes = (unsigned)((P / ((long long)gs->S * gs->Kg)) * gs->N * Cin * 4);
    }
    const int grid = bwd_grid(P, Cout, Cin);
    hipStream_t st = as_stream(stream);
    if (Cout == 128 && Cin == 128) bwd_launch<128, 128>(g, pool, grid, st);
    else if (Cout == 128 && Cin == 96) bwd_launch<128, 96>(g, pool, grid, st);
    else if (Cout == 128 && Cin == 64) bwd_launch<128, 64>(g, pool, grid, st);
    else if (Cout == 96 && Cin == 64) bwd_launch<96, 64>(g, pool, grid, st);
    else bwd_launch<64, 64>(g, pool, grid, st);
    const int n = Cout * Cin;
    hipLaunchKernelGGL(stream_bwd_dw_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, workspace, grid, n, Cin, lddw, dW);
    return prifit_check_launch();
}

int prifit_gemm_stream_bwd_f32(long long P, int Cout, int Cin, const float *G, const float *Y, const float *scale,
                               const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                               const int32_t *pool_arg, const float *pool_T, int pool_K, const float *W, long long ldw,
                               const float *Yp, long long ldyp, const float *p_scale, const float *p_shift,
                               const float *p_mean, const float *p_invstd, float *Gp, long long ldgp, float *red_slab,
                               float *dW, long long lddw, float *workspace, const prifit_bn_bwd *bn, void *stream)
{
    return stream_bwd_impl(P, Cout, Cin, G, Y, scale, shift, coef_a, coef_b, coef_d, pool_arg, pool_T, pool_K, W, ldw, Yp, ldyp,
                           p_scale, p_shift, p_mean, p_invstd, Gp, ldgp, red_slab, dW, lddw, workspace, nullptr, bn, stream);
}

int prifit_gemm_stream_bwd_gather_f32(long long P, int Cout, const float *G, const float *Y, const float *scale,
                                      const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                                      const float *W, long long ldw, const int32_t *idx, const float *U, const float *Vc,
                                      int n_points, int n_centres, int rows_per_centre, const float *p_scale,
                                      const float *p_shift, const float *p_mean, const float *p_invstd, float *Gp, long long ldgp,
                                      float *red_slab, float *dW, long long lddw, float *workspace, const prifit_bn_bwd *bn,
                                      void *stream)
{
    const GatherSrc gs = {idx, U, Vc, n_points, n_centres, rows_per_centre, 64, 0u};
    return stream_bwd_impl(P, Cout, 64, G, Y, scale, shift, coef_a, coef_b, coef_d, nullptr, nullptr, 0, W, ldw, nullptr, 64,
                           p_scale, p_shift, p_mean, p_invstd, Gp, ldgp, red_slab, dW, lddw, workspace, &gs, bn, stream);
}

}  // extern "C"
