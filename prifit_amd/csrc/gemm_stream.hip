// Weights-stationary streaming fp32 GEMM for the tall-and-skinny layers of the shared per-position MLPs
// (models/pointnet_util.py:195-199, :252-256 and their autograd): C[M,N] = A[M,K] . B^T (NT, forward) or
// A[M,K] . B[K,N] (NN, dA) with M = grouped samples (10^5..10^6) and N, K <= 128.
//
// These products are HBM-bound (the A rows are read once, the C rows written once, 2MNK/(4M(K+N)) < 32 flop/B), so
// the kernel is built around bytes in flight rather than around the matrix cores:
//   * persistent workgroups (a few per CU) walk the 64-row tiles of A with a grid stride;
//   * the whole B operand lives in REGISTERS for the life of the workgroup: wave w owns 32 columns of C and keeps
//     its K x 32 slice of B as MFMA fragments, so the k-loop reads only A fragments from LDS and B is fetched once
//     per workgroup instead of once per tile;
//   * the next A tile travels global -> registers while the current one is multiplied; two LDS stages, one barrier
//     per tile; the BatchNorm + ReLU of the producing layer is applied while staging ("normalise on load");
//   * epilogue: bias, stores (two 128-byte row segments per instruction), and per-column sums / sums of squares for
//     the next BatchNorm accumulated in registers over ALL tiles of the workgroup: one statistics slab per workgroup.
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered), same C/D layout as gemm.hip.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SBM = 64;   // rows of A per tile
constexpr int SPAD = 4;   // LDS row padding (floats): conflict-free ds_read_b128 fragments for K = 64, 96, 128

struct StreamArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    long long lda, ldb, ldc;
    const float *a_scale, *a_shift;  // [K] or NULL
    const float *bias;               // [N] or NULL
    float *stats;                    // [gridDim.x][2][N] or NULL
    // RED (dA products): C is the gradient G w.r.t. relu(bn(Yp)) of the previous layer; the epilogue also emits the
    // partial column sums of that layer's BatchNorm backward, m1 = sum(G * mask), m2 = sum(G * mask * yhat) with
    // mask = (Yp * scale + shift > 0), yhat = (Yp - mean) * invstd  (what bn_relu_bwd_reduce would re-read G for)
    const float *red_Y;              // [M, ldry]
    long long ldry;
    const float *red_scale, *red_shift, *red_mean, *red_invstd;  // [N]
    float *red_slab;                 // [gridDim.x][2][N]
    // POOL (dA of the max-pooled last layer): A is the layer's pre-activation Y; the operand dY = b*Y + d + one-hot*T
    // is formed when the fragments are read: b per channel here, d folded into `bias` (d^T W) by the caller, and per
    // pooling group g (pool_K consecutive rows, a multiple of 64) arg[g][c] = winning sample, T[g][c] its gradient
    const int32_t *pool_arg;         // [M / pool_K][K]
    const float *pool_T;             // [M / pool_K][K]
    const float *pool_b;             // [K]
    int pool_K;
    // PMAX (forward of a max-pooled last layer): per 32-row block and column, the largest and the smallest stored C and
    // the row (0..31) of their first occurrence: cand[M / 32][4][N] = (max, argmax, min, argmin; indices as int bits).
    // After the BatchNorm statistics are final, max_k relu(s*y+t) = relu(s*(s >= 0 ? max : min)+t) is read from these
    // candidates instead of from C (prifit_pool_from_candidates).
    float *cand;
    int nslab;                       // slabs the caller reads (stream_grid(M, K)); >= the launched grid
    // BNA (dA of a middle layer): A is the layer's pre-activation Y, bna_G (same leading dimension) the gradient w.r.t.
    // relu(bn(Y)); the operand dY = a * (Y * s + t > 0 ? G : 0) + (b * Y + d) -- bn_relu_bwd_apply's expression -- is
    // formed when the tile is staged (coefficients [K] each)
    const float *bna_G;
    const float *bna_s, *bna_t, *bna_a, *bna_b, *bna_d;
    // GATH (forward, K = 64): A is not stored -- its rows are re-formed from (gs.idx, gs.U, gs.Vc), see GatherSrc
    GatherSrc gs;
    // the column sums (forward statistics, or the RED sums) finalized by this launch instead of written as slabs (common.h)
    BnTail tail;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// WN: waves along N (2 or 4; 256 threads = 4 waves, WM = 4 / WN waves along M); KG = K / 8;
// BKC: B is [N][K] (NT) else [K][N] (NN); AFF: prologue on A
template <int WN, int KG, bool BKC, bool AFF, bool RED, bool POOL, bool PMAX, bool BNA = false, bool GATH = false>
__global__ __launch_bounds__(256, (KG <= 8 ? ((RED || PMAX || POOL) ? 3 : 4) : (KG <= 12 ? ((RED || PMAX || POOL) ? 2 : 3) : 2))) void gemm_stream_kernel(const StreamArgs g)
{
    static_assert(!BNA || (!AFF && !POOL && !PMAX && !BKC), "BNA: a plain NN product with the operand transform");
    static_assert(!GATH || (AFF && BKC && !RED && !POOL && !PMAX && !BNA && (256 % (2 * KG)) == 0), "GATH: the forward product");
    constexpr int K = KG * 8;
    constexpr int WM = 4 / WN;
    constexpr int TM = SBM / (32 * WM);       // 32-row accumulator tiles per wave
    constexpr int LD = K + SPAD;
    constexpr int NV = SBM * K / 4 / 256;     // float4 staged per thread
    static_assert(NV * 256 * 4 == SBM * K, "tile divides evenly");
    __shared__ __attribute__((aligned(16))) float s_a[2][SBM * LD];
    __shared__ __attribute__((aligned(16))) float s_aff[2][K];   // POOL: [0] = b
    __shared__ float s_red[WM][2][32 * WN];
    __shared__ int s_tail;
    __shared__ __attribute__((aligned(16))) float s_bna[BNA ? 5 : 1][BNA ? K : 4];   // s, t, a, b, d

    // (readfirstlane: wm / wn are wave-uniform and the compiler has to know it, see gemm_stream_tn_kernel)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    // (measured, round 4: with N = 96 -- three 32-column blocks for the four waves of WN = 4 -- wave 3 of every workgroup has
    // no columns; rotating the block assignment by the workgroup index, so that the idle wave lands on a different SIMD in
    // each co-resident workgroup, changed nothing: [1.57 M x 96 x 64] 239 us, [1.57 M x 96 x 128 pooled] 504 -> 510 us)
    const int wm = wave / WN, wn = wave % WN;
    const int col = 32 * wn + li;
    const bool col_ok = col < g.N;   // wave-uniform for N % 32 == 0

    if (AFF) {
        for (int t = threadIdx.x; t < K; t += 256) { s_aff[0][t] = g.a_scale[t]; s_aff[1][t] = g.a_shift[t]; }
    }
    if (POOL) {
        for (int t = threadIdx.x; t < K; t += 256) s_aff[0][t] = g.pool_b[t];
    }
    if (BNA) {
        for (int t = threadIdx.x; t < K; t += 256) {
            s_bna[0][t] = g.bna_s[t]; s_bna[1][t] = g.bna_t[t]; s_bna[2][t] = g.bna_a[t]; s_bna[3][t] = g.bna_b[t];
            s_bna[4][t] = g.bna_d[t];
        }
    }
    // this wave's slice of B as fragments: lane (li, lh) of k-group q holds B[col][8q + 4lh + 0..3]
    float4 bf[KG];
#pragma unroll
    for (int q = 0; q < KG; ++q) {
        const int k0 = 8 * q + 4 * lh;
        if (!col_ok) bf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (BKC) bf[q] = ld4(g.B + (long long)col * g.ldb + k0);
        else bf[q] = make_float4(g.B[(long long)k0 * g.ldb + col], g.B[(long long)(k0 + 1) * g.ldb + col],
                                 g.B[(long long)(k0 + 2) * g.ldb + col], g.B[(long long)(k0 + 3) * g.ldb + col]);
    }
    const float bias = (g.bias && col_ok) ? g.bias[col] : 0.f;
    float csum = 0.f, csq = 0.f;
    float r_s = 0.f, r_t = 0.f, r_mu = 0.f, r_is = 0.f, m1 = 0.f, m2 = 0.f;
    if (RED && col_ok) { r_s = g.red_scale[col]; r_t = g.red_shift[col]; r_mu = g.red_mean[col]; r_is = g.red_invstd[col]; }

    const int tiles = (g.M + SBM - 1) / SBM;
    float4 st[NV];
    float4 stg[BNA ? NV : 1];   // BNA: the gradient tile beside the pre-activation tile
    int4 st_arg = make_int4(0, 0, 0, 0);
    float4 st_T = make_float4(0.f, 0.f, 0.f, 0.f);
    // Addressing: one buffer resource per tile (base = the tile's first row, exact extent: rows beyond M read as zeros and
    // stores to them are dropped by the bounds check); a thread's part of an address is a loop-invariant VGPR, the rest
    // is scalar -- no 64-bit VALU address chain per access (the launcher checks SBM * ld * 4 < 2^31).
    int aoff[NV];
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int id = threadIdx.x + 256 * p;
        const int row = id / (K / 4), c4 = id - row * (K / 4);
        aoff[p] = (row * (int)g.lda + 4 * c4) * 4;
    }
    // POOL: the (arg, T) entries of a thread's four channels (the same four for all its NV rows of a tile: K / 4 divides 256)
    static_assert(!POOL || (256 % (K / 4)) == 0, "POOL: one channel group per thread");
    const int pool_c4 = threadIdx.x % (K / 4);
    // GATH: point indices of the thread's NV rows of the tile that is loaded NEXT (requested one tile ahead), the centre
    // term of its four channels
    int nid[GATH ? NV : 1];
    float4 gvc = make_float4(0.f, 0.f, 0.f, 0.f);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(GATH ? g.gs.U : nullptr), 0,
                                                                         GATH ? g.gs.ubytes : 0, 0x00020000);
    auto load_idx = [&](int tile) {
        if (GATH) {
            const int m0 = tile * SBM;
#pragma unroll
            for (int p = 0; p < NV; ++p) nid[p] = g.gs.idx[m0 + (threadIdx.x + 256 * p) / (K / 4)];   // (M % SBM == 0)
        }
    };
    auto tile_rsrc = [&](const float *base, long long ld, int m0, int width) {
        const int rows = g.M - m0 < SBM ? g.M - m0 : SBM;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + (long long)m0 * ld), 0,
                                                 ((rows - 1) * (int)ld + width) * 4, 0x00020000);
    };
    auto load_tile = [&](int tile) {
        const int m0 = tile * SBM;
        // the (arg, T) entries of this tile's pooling group for the thread's four channels, loaded by every thread (no
        // branch) and used when the tile is staged: the operand is formed ONCE per element there instead of once per
        // fragment read (every element is read by all WN waves; on this chip VALU instructions cost matrix time one for one)
        if (POOL) {
            const long long po = (long long)(m0 / g.pool_K) * K + 4 * pool_c4;
            st_arg = *reinterpret_cast<const int4 *>(g.pool_arg + po);
            st_T = ld4(g.pool_T + po);
        }
        if (GATH) {
            // rows of U picked by the tile's index list (read one tile ahead); one centre per tile (Kg % SBM == 0)
            const int grp = m0 / g.gs.Kg;
            const int pbase = (grp / g.gs.S) * g.gs.N;
            gvc = ld4(g.gs.Vc + (long long)grp * K + 4 * pool_c4);
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int n = nid[p];
                const int voff = ((pbase + ((n >= 0 && n < g.gs.N) ? n : 0)) * K + 4 * pool_c4) * 4;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, voff, 0, 0));
                st[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g.A, g.lda, m0, K);
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, aoff[p], 0, 0));
                st[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
        if (BNA) {
            const __amdgpu_buffer_rsrc_t rg = tile_rsrc(g.bna_G, g.lda, m0, K);
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, aoff[p], 0, 0));
                stg[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto store_tile = [&](int tile, float *dst, int stage) {
        const int m0 = tile * SBM;
        const bool full = m0 + SBM <= g.M;   // block-uniform: only the tail tile pays for the row checks
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + 256 * p;
            const int row = id / (K / 4), c4 = id - row * (K / 4);
            float4 x = st[p];
            if (GATH) { x.x -= gvc.x; x.y -= gvc.y; x.z -= gvc.z; x.w -= gvc.w; }   // y = U_j - Vc_g
            if (AFF) {
                const float4 s = *reinterpret_cast<const float4 *>(&s_aff[0][4 * c4]);
                const float4 t = *reinterpret_cast<const float4 *>(&s_aff[1][4 * c4]);
                x.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
                x.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
                if (!full && m0 + row >= g.M) x = make_float4(0.f, 0.f, 0.f, 0.f);   // (without AFF the load returned zeros)
            }
            if (POOL) {   // dY = b * Y + one-hot * T (d is folded into the bias by the caller); a pooling group is a whole number of tiles
                const float4 b4 = *reinterpret_cast<const float4 *>(&s_aff[0][4 * c4]);
                const int kr = m0 % g.pool_K + row;   // sample index of this row in its group
                x.x = fmaf(b4.x, x.x, st_arg.x == kr ? st_T.x : 0.f);
                x.y = fmaf(b4.y, x.y, st_arg.y == kr ? st_T.y : 0.f);
                x.z = fmaf(b4.z, x.z, st_arg.z == kr ? st_T.z : 0.f);
                x.w = fmaf(b4.w, x.w, st_arg.w == kr ? st_T.w : 0.f);
                if (!full && m0 + row >= g.M) x = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (BNA) {   // bn_relu_bwd_apply's expression, term by term
                const float4 cs = *reinterpret_cast<const float4 *>(&s_bna[0][4 * c4]);
                const float4 ct = *reinterpret_cast<const float4 *>(&s_bna[1][4 * c4]);
                const float4 ca = *reinterpret_cast<const float4 *>(&s_bna[2][4 * c4]);
                const float4 cb = *reinterpret_cast<const float4 *>(&s_bna[3][4 * c4]);
                const float4 cd = *reinterpret_cast<const float4 *>(&s_bna[4][4 * c4]);
                const float4 gg = stg[p];
                x.x = fmaf(ca.x, fmaf(x.x, cs.x, ct.x) > 0.f ? gg.x : 0.f, fmaf(cb.x, x.x, cd.x));
                x.y = fmaf(ca.y, fmaf(x.y, cs.y, ct.y) > 0.f ? gg.y : 0.f, fmaf(cb.y, x.y, cd.y));
                x.z = fmaf(ca.z, fmaf(x.z, cs.z, ct.z) > 0.f ? gg.z : 0.f, fmaf(cb.z, x.z, cd.z));
                x.w = fmaf(ca.w, fmaf(x.w, cs.w, ct.w) > 0.f ? gg.w : 0.f, fmaf(cb.w, x.w, cd.w));
                if (!full && m0 + row >= g.M) x = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4 *>(dst + row * LD + 4 * c4) = x;
        }
    };

    const int ldc4 = (int)g.ldc * 4, ldry4 = RED ? (int)g.ldry * 4 : 0;
    const int c_voff = col_ok ? ((wm * 32 * TM + 4 * lh) * (int)g.ldc + col) * 4 : 0x7fffffff;
    const int y_voff = (RED && col_ok) ? ((wm * 32 * TM + 4 * lh) * (int)g.ldry + col) * 4 : 0x7fffffff;
    // Pipeline: the next tile travels global -> registers during the MFMAs of the current one and is staged into the other
    // LDS buffer at the END of the iteration, after the epilogue's stores.  (Staged at the top of the next iteration, the
    // wait for the tile sat at the loop header, where the compiler merges the first iteration's state -- eight loads in
    // flight -- with the back edge's and settles for vmcnt(7): a wait for all but seven of the epilogue's 32 stores, a
    // full write round trip per tile.  Inside one iteration the count is exact.)  Loads and staging are unconditional:
    // the last iteration fetches its own tile again (32 KB per workgroup) instead of branching around the prefetch.
    int tile = blockIdx.x;
    if (tile >= tiles) tile = tiles - 1;   // (the launcher never starts more workgroups than tiles)
    load_idx(tile);
    load_tile(tile);
    {
        const int nx = tile + (int)gridDim.x;
        load_idx(nx < tiles ? nx : tile);
    }
    if (AFF || POOL || BNA) __syncthreads();  // s_aff / s_bna visible before the first staging
    store_tile(tile, s_a[0], 0);
    __syncthreads();
    for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
        float *As = s_a[it & 1];
        const int next = tile + gridDim.x;
        const int ntile = next < tiles ? next : tile;
        load_tile(ntile);
        if (GATH) {   // the index list of the tile after that, a whole iteration ahead of the loads that use it
            const int n2 = ntile + (int)gridDim.x;
            load_idx(n2 < tiles ? n2 : ntile);
        }

        f32x16 acc[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        // RED: the previous layer's pre-activations under this wave's output elements, in flight during the MFMAs
        float yp[TM][16];
        if (RED) {
            const __amdgpu_buffer_rsrc_t yrs = tile_rsrc(g.red_Y, g.ldry, tile * SBM, g.N);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yp[a][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        yrs, y_voff, (32 * a + (r & 3) + 8 * (r >> 2)) * ldry4, 0));
        }
        // the prefetch (and the RED loads) are issued HERE, ahead of the MFMAs: left alone, the scheduler sinks them behind
        // the matrix instructions (fewer live registers) and their latency is exposed in the epilogue
        __builtin_amdgcn_sched_barrier(0);
        const float *ap = As + (wm * 32 * TM + li) * LD + 4 * lh;
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            float4 fa[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const float4 *>(ap + a * 32 * LD + 8 * q);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, bf[q].x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, bf[q].y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, bf[q].z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, bf[q].w, acc[a], 0, 0, 0);
            }
        }
        // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        if (col_ok) {   // wave-uniform
            const int m0 = tile * SBM + wm * 32 * TM + 4 * lh;
            const bool full = tile * SBM + SBM <= g.M;
            const __amdgpu_buffer_rsrc_t crs = tile_rsrc(g.C, g.ldc, tile * SBM, g.N);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float vmax = -INFINITY, vmin = INFINITY;
                int imax = 0, imin = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * a + (r & 3) + 8 * (r >> 2);
                    const bool rok = full || row < g.M;   // (tail tile only; its stores are dropped by the bounds check)
                    const float v = acc[a][r] + bias;
                    const float vs = rok ? v : 0.f;
                    csum += vs;
                    csq += vs * vs;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, c_voff,
                                                          (32 * a + (r & 3) + 8 * (r >> 2)) * ldc4, 0);
                    if (PMAX) {  // rows ascend with r inside a lane: strict comparisons keep the first occurrence
                        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (rok && v > vmax) { vmax = v; imax = ri; }
                        if (rok && v < vmin) { vmin = v; imin = ri; }
                    }
                    if (RED) {
                        const float y = yp[a][r];
                        const float gm = fmaf(y, r_s, r_t) > 0.f ? vs : 0.f;
                        m1 += gm;
                        m2 += gm * ((y - r_mu) * r_is);
                    }
                }
                if (PMAX) {
                    // the other half of the block's rows lives in lane ^ 32; ties go to the lower row
                    const float ov = __shfl_xor(vmax, 32, 64), on = __shfl_xor(vmin, 32, 64);
                    const int oi = __shfl_xor(imax, 32, 64), oj = __shfl_xor(imin, 32, 64);
                    if (ov > vmax || (ov == vmax && oi < imax)) { vmax = ov; imax = oi; }
                    if (on < vmin || (on == vmin && oj < imin)) { vmin = on; imin = oj; }
                    const int blk_row = tile * SBM + wm * 32 * TM + 32 * a;
                    if (lh == 0 && blk_row < g.M) {
                        float *cd = g.cand + (long long)(blk_row >> 5) * 4 * g.N + col;
                        cd[0] = vmax; cd[g.N] = __int_as_float(imax); cd[2 * g.N] = vmin; cd[3 * g.N] = __int_as_float(imin);
                    }
                }
            }
        }
        store_tile(ntile, s_a[(it + 1) & 1], (it + 1) & 1);
        __syncthreads();
    }
    if (RED) { csum = m1; csq = m2; }
    float *slab_out = RED ? g.red_slab : g.stats;
    const bool tail = g.tail.acc != nullptr;
    if (slab_out || tail) {
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (lh == 0) { s_red[wm][0][32 * wn + li] = csum; s_red[wm][1][32 * wn + li] = csq; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * g.N; t += 256) {
            const int which = t / g.N, c = t - which * g.N;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += s_red[w][which][c];
            if (tail) { bn_tail_add(g.tail, which, c, s); continue; }
            slab_out[((long long)blockIdx.x * 2 + which) * g.N + c] = s;
            // the caller sums g.nslab slabs (prifit_gemm_stream_slabs); this variant may run on fewer workgroups
            for (int extra = blockIdx.x + gridDim.x; extra < g.nslab; extra += gridDim.x)
                slab_out[((long long)extra * 2 + which) * g.N + c] = 0.f;
        }
        if (tail) bn_tail_finish(g.tail, &s_tail);
    }
}

// dW = G^T . relu(bn(A)) over the grouped samples: out[Mo, No] += sum_rows G[row, 0:Mo]^T A[row, 0:No]  (TN, the
// reduction runs over 10^5..10^6 rows, the output is at most 128 x 128).  Both operands stream; there is nothing to
// share between tiles, so there is no LDS and no barrier at all: every WAVE walks its own rows and loads the MFMA
// fragments straight from global memory (lane = output row / column, 2 x 128 contiguous bytes per load instruction).
//   * AM x AN (32 x 32 tiles of the output per wave) is chosen by the launcher so that the tiles divide EVENLY over
//     the four waves (128 x 96 = 4 waves of 1 x 3 tiles, 96 x 64 = 2 sub-blocks of 3 x 1 tiles x 2 row streams, ...):
//     every SIMD gets the same number of matrix instructions per row group;
//   * PF 8-row groups per wave are in flight: the loop is unrolled PF times over statically indexed register
//     stages (a rotating copy would wait for the newest load at the end of every iteration), loads are
//     unconditional (rows past the end of the wave's range are clamped to its last group and multiplied by zero),
//     so the steady state is straight-line code with exact vmcnt waits;
//   * the row-group streams of a workgroup are combined through LDS, every workgroup stores one partial [Mo, No]
//     slab into the caller's workspace (plain stores: hundreds of workgroups hammering the same few KB with atomics
//     cost more than the streaming itself), and a second tiny launch adds the slabs to `out`.
struct StreamTNArgs {
    const float *G, *A;
    float *ws;   // [gridDim.x][Mo * No] partial slabs
    int Mo, No;
    long long P, ldg, lda, rows_per_wg;
    const float *b_scale, *b_shift;  // prologue on A (channel = output column) or NULL
    // POOL: G is the pooled layer's pre-activation Y and the operand dY = b*Y + d + one-hot*T is formed on load
    // (lane = channel: b, d are per-lane constants; arg / T rows of the pooling group once per 8-row group)
    const int32_t *pool_arg;         // [P / pool_K][Mo]
    const float *pool_T;             // [P / pool_K][Mo]
    const float *pool_b, *pool_d;    // [Mo]
    int pool_K;                      // multiple of 8
    int interleave;
    // BNA (dW of a middle layer): G is the layer's pre-activation Y and bn_G the gradient w.r.t. relu(bn(Y)); the operand
    // dY = a * (Y * s + t > 0 ? bn_G : 0) + (b * Y + d) -- exactly bn_relu_bwd_apply's expression -- is formed on load
    // (lane = channel: s, t, a, b, d are per-lane constants; bn_G shares Y's leading dimension)
    const float *bn_G;
    const float *bn_s, *bn_t, *bn_a, *bn_b, *bn_d;   // [Mo]
};
#ifndef TN_DEBUG
#define TN_DEBUG 0
#endif

// RAG: the output is not a whole number of AM x AN sub-blocks (96 x 96 only): tiles beyond it are skipped at run time
template <int AM, int AN, int PF, int OCC, bool AFF, bool POOL, bool RAG, bool BNA = false>
__global__ __launch_bounds__(256, OCC) void gemm_stream_tn_kernel(const StreamTNArgs g)
{
    static_assert(!(POOL && BNA), "one operand transform at a time");
    constexpr int NT = AM * AN;
    __shared__ float s_part[3][NT * 1024];  // partial sub-blocks of the row-group streams ks = 1..3
    // (readfirstlane: everything derived from the wave index is wave-uniform, and the compiler has to know it -- as a
    // "divergent" value it wrapped every load and every MFMA in its own exec-mask branch and waited vmcnt(0) in the loop)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int mt_total = g.Mo >> 5, nt_total = g.No >> 5;
    const int mb = (mt_total + AM - 1) / AM, nb = (nt_total + AN - 1) / AN;
    const int SB = mb * nb, KS = 4 / SB;          // sub-blocks (1, 2 or 4) x interleaved row-group streams
    const int sb = wave % SB, ks = wave / SB;
    const int mblk = sb / nb, nblk = sb - mblk * nb;
    const int m0 = 32 * AM * mblk, n0 = 32 * AN * nblk;
    const int m_cnt = RAG ? min(AM, mt_total - AM * mblk) : AM, n_cnt = RAG ? min(AN, nt_total - AN * nblk) : AN;

    f32x16 acc[AM][AN];
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < AN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float sc[AN], sh[AN];
#pragma unroll
    for (int b = 0; b < AN; ++b) {
        sc[b] = 1.f; sh[b] = 0.f;
        if (AFF && b < n_cnt) { sc[b] = g.b_scale[n0 + 32 * b + li]; sh[b] = g.b_shift[n0 + 32 * b + li]; }
    }
    float pb[AM], pd[AM];
    float ns[AM], nt[AM], na[AM];   // BNA: scale, shift, a (b and d share pb / pd)
#pragma unroll
    for (int a = 0; a < AM; ++a) {
        pb[a] = 0.f; pd[a] = 0.f; ns[a] = 0.f; nt[a] = 0.f; na[a] = 0.f;
        if (POOL && a < m_cnt) { pb[a] = g.pool_b[m0 + 32 * a + li]; pd[a] = g.pool_d[m0 + 32 * a + li]; }
        if (BNA && a < m_cnt) {
            const int c = m0 + 32 * a + li;
            pb[a] = g.bn_b[c]; pd[a] = g.bn_d[c]; ns[a] = g.bn_s[c]; nt[a] = g.bn_t[c]; na[a] = g.bn_a[c];
        }
    }
    // rows of this wave's stream.  interleave: the 8-row groups of the whole launch are dealt round-robin over all
    // (workgroup, stream) pairs, so at any moment the chip reads ONE contiguous window of the operands; otherwise every
    // workgroup walks its own contiguous range.  Everything about a row group is wave-uniform and lives in SGPRs.
    const int P = (int)g.P, rpw = (int)g.rows_per_wg;
    const int r0 = g.interleave ? 0 : blockIdx.x * rpw;
    const int r1 = g.interleave ? P : (r0 + rpw < P ? r0 + rpw : P);  // r1 - r0: a positive multiple of 8
    const int stride = g.interleave ? 8 * KS * (int)gridDim.x : 8 * KS;
    // Addressing: one buffer resource per 8-row group (base = the group's first row, a scalar 64-bit add), the lane's
    // part of the address is ONE loop-invariant VGPR per operand, the row inside the group a loop-invariant SGPR offset:
    // no vector instruction per load at all.  (With flat loads every load carried its own 64-bit VALU address chain;
    // the waves of a SIMD fell into step -- all computing addresses, then all queueing for the matrix pipe -- and the
    // MFMA rate of this kernel was ~0.55 of peak even with every load hitting the cache.)
    const int ldg4 = (int)g.ldg * 4, lda4 = (int)g.lda * 4;
    const int g_voff = (4 * lh * (int)g.ldg + m0 + li) * 4, a_voff = (4 * lh * (int)g.lda + n0 + li) * 4;
    const int g_bytes = (7 * (int)g.ldg + g.Mo) * 4, a_bytes = (7 * (int)g.lda + g.No) * 4;
    const int p_voff = (m0 + li) * 4;
    const __amdgpu_buffer_rsrc_t arg_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t *>(POOL ? g.pool_arg : nullptr), 0, POOL ? (P / g.pool_K) * g.Mo * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t t_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(POOL ? g.pool_T : nullptr), 0, POOL ? (P / g.pool_K) * g.Mo * 4 : 0, 0x00020000);
    // pooling group (quotient) and sample index (remainder) of a row advance incrementally: no division in the loop
    const int pk = POOL ? g.pool_K : 1;
    const int sq = stride / pk, sr = stride - sq * pk, q_last = (r1 - 8) / pk;

    float fg[PF][AM][4], fa[PF][AN][4];
    float fgg[PF][BNA ? AM : 1][4];   // BNA: the gradient fragments beside the pre-activation fragments
    int fw[PF][AM];      // POOL: winning sample of this lane's channel in the row group's pool
    float ft[PF][AM];    //       and its gradient
    auto load = [&](int row_, int q_, float (&xg)[AM][4], float (&xa)[AN][4], int (&xw)[AM], float (&xt)[AM], float (&xgg)[BNA ? AM : 1][4]) {
        const bool live = row_ < r1;
        const int row = live ? row_ : r1 - 8;   // past the end: any valid group (its product is zeroed)
        const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.G + (long long)row * g.ldg), 0, g_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.A + (long long)row * g.lda), 0, a_bytes, 0x00020000);
        if (POOL) {
            const int so = (live ? q_ : q_last) * g.Mo * 4;
#pragma unroll
            for (int a = 0; a < AM; ++a) {
                xw[a] = 0; xt[a] = 0.f;
                if (a < m_cnt) {
                    xw[a] = (int)__builtin_amdgcn_raw_buffer_load_b32(arg_rs, p_voff + 128 * a, so, 0);
                    xt[a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t_rs, p_voff + 128 * a, so, 0));
                }
            }
        }
#pragma unroll
        for (int a = 0; a < AM; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xg[a][j] = 0.f;
            if (a < m_cnt) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xg[a][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, g_voff + 128 * a, j * ldg4, 0));
            }
        }
        if (BNA) {
            const __amdgpu_buffer_rsrc_t ggrs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(g.bn_G + (long long)row * g.ldg), 0, g_bytes, 0x00020000);
#pragma unroll
            for (int a = 0; a < AM; ++a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xgg[a][j] = 0.f;
                if (a < m_cnt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xgg[a][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ggrs, g_voff + 128 * a, j * ldg4, 0));
                }
            }
        }
#pragma unroll
        for (int b = 0; b < AN; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[b][j] = 0.f;
            if (b < n_cnt) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xa[b][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, a_voff + 128 * b, j * lda4, 0));
            }
        }
    };
    auto multiply = [&](int row, int rem, float (&xg)[AM][4], float (&xa)[AN][4], const int (&xw)[AM], const float (&xt)[AM],
                        const float (&xgg)[BNA ? AM : 1][4]) {
        const bool live = row < r1;   // wave-uniform
        if (BNA) {   // bn_relu_bwd_apply's expression, term by term
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xg[a][j] = fmaf(na[a], fmaf(xg[a][j], ns[a], nt[a]) > 0.f ? xgg[a][j] : 0.f, fmaf(pb[a], xg[a][j], pd[a]));
        }
        if (POOL) {
            const int kb = rem + 4 * lh;  // sample index of this lane's first row in its pooling group
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) xg[a][j] = fmaf(pb[a], xg[a][j], pd[a]) + (xw[a] == kb + j ? xt[a] : 0.f);   // (pool_bwd_apply's order of additions)
        }
        if (!live) {   // (a scalar branch: nothing of it in the steady state -- VALU instructions cost matrix time here)
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) xg[a][j] = 0.f;
        }
        if (AFF) {
#pragma unroll
            for (int b = 0; b < AN; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) xa[b][j] = fmaxf(fmaf(xa[b][j], sc[b], sh[b]), 0.f);
        }
        if (TN_DEBUG == 1) {   // DIAGNOSIS: memory side only (one VALU op per fragment pair keeps the loads alive)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < AM; ++a)
#pragma unroll
                    for (int b = 0; b < AN; ++b) acc[a][b][0] = fmaf(xg[a][j], xa[b][j], acc[a][b][0]);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int b = 0; b < AN; ++b)
                    if (a < m_cnt && b < n_cnt)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xg[a][j], xa[b][j], acc[a][b], 0, 0, 0);
    };
    int row = g.interleave ? 8 * ((int)blockIdx.x * KS + ks) : r0 + 8 * ks;   // the group being multiplied
    int mrem = row % pk;
    int lrow = row, lq = row / pk, lrem = mrem;                                // the group being requested
    auto advance = [&](int &r, int &q, int &rem) {
        r += stride; q += sq; rem += sr;
        if (rem >= pk) { rem -= pk; ++q; }
    };
    if (TN_DEBUG == 2) { lrow = 8 * wave; lq = 0; }   // DIAGNOSIS: matrix side only (every load hits the same lines)
#pragma unroll
    for (int s = 0; s < PF - 1; ++s) {
        load(lrow, lq, fg[s], fa[s], fw[s], ft[s], fgg[s]);
        if (TN_DEBUG != 2) advance(lrow, lq, lrem);
    }
    while (row < r1) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = (s + PF - 1) % PF;
            load(lrow, lq, fg[t], fa[t], fw[t], ft[t], fgg[t]);
            if (TN_DEBUG != 2) advance(lrow, lq, lrem);
            // the loads stay HERE, a whole PF - 1 groups ahead of their use: left alone, the scheduler sinks each one to
            // just before its MFMA (fewer live registers) and the wave waits out the memory latency every time
            __builtin_amdgcn_sched_barrier(0);
            multiply(row, mrem, fg[s], fa[s], fw[s], ft[s], fgg[s]);
            __builtin_amdgcn_sched_barrier(0);
            int dummy_q = 0;
            advance(row, dummy_q, mrem);
        }
    }
    // combine the KS row-group streams of every sub-block through LDS (stream 0 of each sub-block collects)
    if (KS > 1) {
        if (ks > 0) {
            float *dst = s_part[wave - SB];
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int b = 0; b < AN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((a * AN + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (ks > 0) return;
        for (int k = 1; k < KS; ++k) {
            const float *src = s_part[k * SB + sb - SB];
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int b = 0; b < AN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * AN + b) * 16 + r) * 64 + lane];
        }
    }
    // C/D layout: col = lane & 31 (n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (m)
    float *slab = g.ws + (long long)blockIdx.x * g.Mo * g.No;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
        if (a >= m_cnt) continue;
#pragma unroll
        for (int b = 0; b < AN; ++b) {
            if (b >= n_cnt) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[m * g.No + n0 + 32 * b + li] = acc[a][b][r];
            }
        }
    }
}

// out[m, n] += sum over slabs of ws[slab][m * No + n]; gridDim.y slices of the slab range (32 adders per address), four
// independent partial sums per thread so that the slab loads of a slice are in flight together
__global__ __launch_bounds__(256) void stream_tn_reduce_kernel(const float *__restrict__ ws, int nslab, int Mo, int No,
                                                               long long ldo, float *__restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Mo * No) return;
    const int per = (nslab + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * per, s1 = min(nslab, s0 + per);
    const long long stride = (long long)Mo * No;
    const float *p = ws + (long long)s0 * stride + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sl = s0;
    for (; sl + 4 <= s1; sl += 4, p += 4 * stride) {
        a0 += p[0]; a1 += p[stride]; a2 += p[2 * stride]; a3 += p[3 * stride];
    }
    for (; sl < s1; ++sl, p += stride) a0 += p[0];
    const int m = e / No, n = e - m * No;
    if (s1 > s0) unsafeAtomicAdd(out + (long long)m * ldo + n, (a0 + a1) + (a2 + a3));
}

int stream_grid(int M, int K)
{
    const int tiles = (M + SBM - 1) / SBM;
    const int cap = 256 * (K <= 64 ? 4 : (K <= 96 ? 3 : 2));   // CUs x resident workgroups (LDS: 2 stages of 64 x (K+4) floats)
    return tiles < cap ? tiles : cap;
}

template <int WN, int KG, bool BKC>
void launch_aff(const StreamArgs &g_, int nslab, hipStream_t st)
{
    // persistent grid = the workgroups that are resident at once for THIS variant (its __launch_bounds__; a grid sized for
    // the plain variant put the register-heavy ones through 1.33 - 1.5 rounds); the statistics slabs beyond it are zeroed
    StreamArgs g = g_;
    g.nslab = nslab;
    const bool red = g.red_slab || g.tail.kind == 2;
    const bool heavy = g.pool_arg || red || g.cand || g.bna_G;
    // (round 6: the pool-candidate forward at K = 96 uses 92 VGPRs and 52 KB of LDS, so a third workgroup per CU fits -- measured
    // 432 -> 455 us for [1.57 M x 128 x 96]: more resident workgroups only add to the queue in front of the memory system)
    const int occ = KG <= 8 ? (heavy ? 3 : 4) : (KG <= 12 ? (heavy ? 2 : 3) : 2);
    const int grid = nslab < 256 * occ ? nslab : 256 * occ;
    if (!BKC && g.bna_G && red) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, true, false, false, true>), dim3(grid), dim3(256), 0, st, g);
    else if (!BKC && g.pool_arg) {
        // (K = 96 pooled layers do not exist: a thread's rows of a tile would not share their four channels; launch_k refuses them)
        if constexpr (256 % (2 * KG) == 0) {
            if (red) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, true, true, false>), dim3(grid), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, false, true, false>), dim3(grid), dim3(256), 0, st, g);
        }
    }
    else if (!BKC && red) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, true, false, false>), dim3(grid), dim3(256), 0, st, g);
    else if (BKC && g.gs.idx) {
        if constexpr (KG == 8 && BKC) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, true, true, false, false, false, false, true>), dim3(grid), dim3(256), 0, st, g);
    }
    else if (BKC && g.cand && g.a_scale) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, true, true, false, false, true>), dim3(grid), dim3(256), 0, st, g);
    else if (g.a_scale) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, BKC, true, false, false, false>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, BKC, false, false, false, false>), dim3(grid), dim3(256), 0, st, g);
}

template <int WN, bool BKC>
int launch_k(const StreamArgs &g, int grid, hipStream_t st)
{
    switch (g.K) {
        case 64: launch_aff<WN, 8, BKC>(g, grid, st); break;
        case 96:
            if (g.pool_arg) return PRIFIT_EINVAL;
            launch_aff<WN, 12, BKC>(g, grid, st);
            break;
        case 128: launch_aff<WN, 16, BKC>(g, grid, st); break;
        default: return PRIFIT_EINVAL;
    }
    return prifit_check_launch();
}

}  // namespace

// (AM, AN) tiles per wave, prefetch depth and resident workgroups per CU for an [Mo, No] output: AM x AN <= 4 tiles
// dividing the output into 1, 2 or 4 sub-blocks (only 96 x 96 has no such split: ragged 2 x 2)
struct TNPlan { int am, an, pf, occ, ragged; };
static TNPlan stream_tn_plan(int Mo, int No)
{
    const int mt = Mo >> 5, nt = No >> 5;
    TNPlan p;
    p.ragged = 0;
    if (mt == 3 && nt == 3) { p.am = 2; p.an = 2; p.ragged = 1; }
    else if (mt == 3) { p.am = 3; p.an = 1; }
    else if (nt == 3) { p.am = 1; p.an = 3; }
    else { p.am = mt >= 2 ? 2 : 1; p.an = nt >= 2 ? 2 : 1; }
    p.pf = 3;   // register stages in flight (2 and 4 measured slower)
    p.occ = 3;
    return p;
}

static long long stream_tn_split(int Mo, int No, long long P, long long *per_out)
{
    long long nwg = 256 * stream_tn_plan(Mo, No).occ;   // the resident workgroups
    if (nwg > P / 256) nwg = P / 256 > 0 ? P / 256 : 1; // short reductions: fewer, longer ranges (fewer slabs to add up)
    long long per = (P + nwg - 1) / nwg;
    per = (per + 31) / 32 * 32;                 // whole 8-row groups for every interleaved stream
    *per_out = per;
    return (P + per - 1) / per;
}

template <int AM, int AN, int PF, int OCC, bool RAG>
static void stream_tn_launch_t(const StreamTNArgs &g, dim3 grid, hipStream_t st)
{
    const dim3 block(256);
    if (g.bn_G) {   // (a middle layer always has the prologue on A)
        hipLaunchKernelGGL((gemm_stream_tn_kernel<AM, AN, PF, OCC, true, false, RAG, true>), grid, block, 0, st, g);
        return;
    }
    if (g.pool_arg) {
        if (g.b_scale) hipLaunchKernelGGL((gemm_stream_tn_kernel<AM, AN, PF, OCC, true, true, RAG>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_stream_tn_kernel<AM, AN, PF, OCC, false, true, RAG>), grid, block, 0, st, g);
    } else {
        if (g.b_scale) hipLaunchKernelGGL((gemm_stream_tn_kernel<AM, AN, PF, OCC, true, false, RAG>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_stream_tn_kernel<AM, AN, PF, OCC, false, false, RAG>), grid, block, 0, st, g);
    }
}

template <int AM, int AN, bool RAG = false>
static void stream_tn_launch_pf(const StreamTNArgs &g, const TNPlan &p, dim3 grid, hipStream_t st)
{
    // (the BatchNorm-apply variant of the 2 x 2 plan spills two registers at depth 3)
    if (g.bn_G && AM * AN == 4) stream_tn_launch_t<AM, AN, 2, 3, RAG>(g, grid, st);
    else stream_tn_launch_t<AM, AN, 3, 3, RAG>(g, grid, st);
}

static int stream_tn_launch(StreamTNArgs &g, float *out, long long ldo, void *stream)
{
    const long long nwg = stream_tn_split(g.Mo, g.No, g.P, &g.rows_per_wg);
    const TNPlan p = stream_tn_plan(g.Mo, g.No);
    g.interleave = 1;
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)nwg);
    if (p.ragged) stream_tn_launch_pf<2, 2, true>(g, p, grid, st);
    else if (p.am == 3) stream_tn_launch_pf<3, 1>(g, p, grid, st);
    else if (p.an == 3) stream_tn_launch_pf<1, 3>(g, p, grid, st);
    else if (p.am == 2 && p.an == 2) stream_tn_launch_pf<2, 2>(g, p, grid, st);
    else if (p.am == 2) stream_tn_launch_pf<2, 1>(g, p, grid, st);
    else if (p.an == 2) stream_tn_launch_pf<1, 2>(g, p, grid, st);
    else stream_tn_launch_pf<1, 1>(g, p, grid, st);
    hipLaunchKernelGGL(stream_tn_reduce_kernel, dim3((g.Mo * g.No + 255) / 256, nwg < 32 ? (unsigned)nwg : 32u), dim3(256), 0, st,
                       g.ws, (int)nwg, g.Mo, g.No, ldo, out);
    return prifit_check_launch();
}

extern "C" {

int prifit_gemm_stream_supported(int layout, int M, int N, int K)
{
    // (M < 2^24: the tile loads use 32-bit element offsets, M * lda < 2^31 with lda <= 128 .. checked again per call)
    return (layout == 0 || layout == 1) && M >= 32768 && M < (1 << 24) && (N == 64 || N == 96 || N == 128) &&
           (K == 64 || K == 96 || K == 128);
}

int prifit_gemm_stream_slabs(int M, int K) { return stream_grid(M, K); }

static int stream_launch(StreamArgs &g, int layout, void *stream)
{
    // 32-bit byte offsets inside a 64-row tile
    if (g.lda >= (1 << 22) || g.ldc >= (1 << 22) || (g.red_Y && g.ldry >= (1 << 22))) return PRIFIT_EINVAL;
    const int grid = stream_grid(g.M, g.K);
    hipStream_t st = as_stream(stream);
    if (g.N == 64) return layout == 0 ? launch_k<2, true>(g, grid, st) : launch_k<2, false>(g, grid, st);
    return layout == 0 ? launch_k<4, true>(g, grid, st) : launch_k<4, false>(g, grid, st);
}

int prifit_gemm_stream_dgrad_f32(int M, int N, int K, const float *dY, long long lda, const float *W, long long ldb,
                                 float *G, long long ldc, const float *Yprev, long long ldy, const float *scale,
                                 const float *shift, const float *mean, const float *invstd, float *red_slab,
                                 const prifit_bn_bwd *bn, void *stream)
{
    if (!dY || !W || !G || !Yprev || !scale || !shift || !mean || !invstd || (!red_slab && !(bn && bn->acc)) || bn_bwd_bad(bn) ||
        !prifit_gemm_stream_supported(1, M, N, K) || lda < K || ldc < N || ldb < N || ldy < N || (lda & 3) ||
        ((uintptr_t)dY & 15))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.gs.idx = nullptr;
    g.A = dY; g.B = W; g.C = G; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = nullptr; g.a_shift = nullptr; g.bias = nullptr; g.stats = nullptr;
    g.red_Y = Yprev; g.ldry = ldy; g.red_scale = scale; g.red_shift = shift; g.red_mean = mean; g.red_invstd = invstd;
    g.red_slab = red_slab;
    g.tail = bn_tail_bwd(bn, N);
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = nullptr; g.bna_G = nullptr;
    return stream_launch(g, 1, stream);
}

int prifit_gemm_stream_dgrad_bn_f32(int M, int N, int K, const float *Gin, const float *Y, long long lda, const float *W,
                                    long long ldb, float *G, long long ldc, const float *scale_l, const float *shift_l,
                                    const float *coef_a, const float *coef_b, const float *coef_d, const float *Yprev,
                                    long long ldy, const float *scale, const float *shift, const float *mean,
                                    const float *invstd, float *red_slab, const prifit_bn_bwd *bn, void *stream)
{
    if (!Gin || !Y || !W || !G || !scale_l || !shift_l || !coef_a || !coef_b || !coef_d || !Yprev || !scale || !shift ||
        !mean || !invstd || (!red_slab && !(bn && bn->acc)) || bn_bwd_bad(bn) || !prifit_gemm_stream_supported(1, M, N, K) || lda < K || ldc < N || ldb < N ||
        ldy < N || (lda & 3) || ((uintptr_t)Y & 15) || ((uintptr_t)Gin & 15))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.gs.idx = nullptr;
    g.A = Y; g.B = W; g.C = G; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = nullptr; g.a_shift = nullptr; g.bias = nullptr; g.stats = nullptr;
    g.red_Y = Yprev; g.ldry = ldy; g.red_scale = scale; g.red_shift = shift; g.red_mean = mean; g.red_invstd = invstd;
    g.red_slab = red_slab;
    g.tail = bn_tail_bwd(bn, N);
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = nullptr;
    g.bna_G = Gin; g.bna_s = scale_l; g.bna_t = shift_l; g.bna_a = coef_a; g.bna_b = coef_b; g.bna_d = coef_d;
    return stream_launch(g, 1, stream);
}

int prifit_gemm_stream_dgrad_pool_f32(int M, int N, int K, const float *Y, long long lda, const float *W, long long ldb,
                                      float *G, long long ldc, const float *bias_dW, const int32_t *pool_arg,
                                      const float *pool_T, const float *coef_b, int pool_K, const float *Yprev,
                                      long long ldy, const float *scale, const float *shift, const float *mean,
                                      const float *invstd, float *red_slab, const prifit_bn_bwd *bn, void *stream)
{
    if (bn_bwd_bad(bn)) return PRIFIT_EINVAL;
    if (!Y || !W || !G || !pool_arg || !pool_T || !coef_b || !prifit_gemm_stream_supported(1, M, N, K) || lda < K ||
        ldc < N || ldb < N || (lda & 3) || ((uintptr_t)Y & 15) || pool_K < 64 || (pool_K & 63) || (M % pool_K) ||
        (((uintptr_t)pool_arg | (uintptr_t)pool_T) & 15))
        return PRIFIT_EINVAL;
    if ((red_slab || (bn && bn->acc)) && (!Yprev || !scale || !shift || !mean || !invstd || ldy < N)) return PRIFIT_EINVAL;
    StreamArgs g;
    g.gs.idx = nullptr;
    g.A = Y; g.B = W; g.C = G; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = nullptr; g.a_shift = nullptr; g.bias = bias_dW; g.stats = nullptr;
    g.red_Y = Yprev; g.ldry = ldy; g.red_scale = scale; g.red_shift = shift; g.red_mean = mean; g.red_invstd = invstd;
    g.red_slab = red_slab;
    g.tail = bn_tail_bwd(bn, N);
    g.pool_arg = pool_arg; g.pool_T = pool_T; g.pool_b = coef_b; g.pool_K = pool_K; g.cand = nullptr; g.bna_G = nullptr;
    return stream_launch(g, 1, stream);
}

int prifit_gemm_stream_f32(int layout, int M, int N, int K, const float *A, long long lda, const float *B,
                           long long ldb, float *C, long long ldc, const float *a_scale, const float *a_shift,
                           const float *bias, float *col_stats, const prifit_bn_fwd *bn, void *stream)
{
    if (bn_fwd_bad(bn)) return PRIFIT_EINVAL;
    if (!A || !B || !C || !prifit_gemm_stream_supported(layout, M, N, K) || lda < K || ldc < N ||
        ((a_scale == nullptr) != (a_shift == nullptr)) || (lda & 3) || ((uintptr_t)A & 15) ||
        (layout == 0 && ((ldb & 3) || ((uintptr_t)B & 15) || ldb < K)) || (layout == 1 && ldb < N))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.gs.idx = nullptr;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = a_scale; g.a_shift = a_shift; g.bias = bias; g.stats = col_stats;
    g.red_Y = nullptr; g.ldry = 0; g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr; g.red_slab = nullptr;
    g.tail = bn_tail_fwd(bn, N);
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = nullptr; g.bna_G = nullptr;
    return stream_launch(g, layout, stream);
}

static bool gather_src_ok(const GatherSrc &gs, long long P, int C)
{
    return gs.idx && gs.U && gs.Vc && gs.N > 0 && gs.S > 0 && gs.Kg > 0 && (gs.Kg % SBM) == 0 && gs.C == C &&
           P % ((long long)gs.S * gs.Kg) == 0 && !(((uintptr_t)gs.U | (uintptr_t)gs.Vc) & 15) &&
           (P / ((long long)gs.S * gs.Kg)) * gs.N * C * 4 < 0x7ff00000LL;
}

int prifit_gemm_stream_gather_f32(int M, int N, const int32_t *idx, const float *U, const float *Vc, int n_points, int n_centres,
                                  int rows_per_centre, const float *B, long long ldb, float *C, long long ldc,
                                  const float *a_scale, const float *a_shift, const float *bias, float *col_stats,
                                  const prifit_bn_fwd *bn, void *stream)
{
    const int K = 64;
    if (bn_fwd_bad(bn)) return PRIFIT_EINVAL;
    GatherSrc gs = {idx, U, Vc, n_points, n_centres, rows_per_centre, K, 0u};
    if (!B || !C || !a_scale || !a_shift || !prifit_gemm_stream_supported(0, M, N, K) || ldc < N || (ldb & 3) ||
        ((uintptr_t)B & 15) || ldb < K || !gather_src_ok(gs, M, K))
        return PRIFIT_EINVAL;
    gs.ubytes = (unsigned)((M / ((long long)n_centres * rows_per_centre)) * n_points * K * 4);
    StreamArgs g;
    g.A = U; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = a_scale; g.a_shift = a_shift; g.bias = bias; g.stats = col_stats;
    g.red_Y = nullptr; g.ldry = 0; g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr; g.red_slab = nullptr;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = nullptr; g.bna_G = nullptr;
    g.tail = bn_tail_fwd(bn, N);
    g.gs = gs;
    return stream_launch(g, 0, stream);
}

int prifit_gemm_stream_pool_f32(int M, int N, int K, const float *A, long long lda, const float *B, long long ldb,
                                float *C, long long ldc, const float *a_scale, const float *a_shift, const float *bias,
                                float *col_stats, float *cand, const prifit_bn_fwd *bn, void *stream)
{
    if (bn_fwd_bad(bn)) return PRIFIT_EINVAL;
    if (!A || !B || !C || !cand || !a_scale || !a_shift || !prifit_gemm_stream_supported(0, M, N, K) || lda < K ||
        ldc < N || (lda & 3) || ((uintptr_t)A & 15) || (ldb & 3) || ((uintptr_t)B & 15) || ldb < K || (M & 31))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.gs.idx = nullptr;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = a_scale; g.a_shift = a_shift; g.bias = bias; g.stats = col_stats;
    g.red_Y = nullptr; g.ldry = 0; g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr; g.red_slab = nullptr;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = cand; g.bna_G = nullptr;
    g.tail = bn_tail_fwd(bn, N);
    return stream_launch(g, 0, stream);
}

int prifit_gemm_stream_tn_supported(int Mo, int No, long long P)
{
    return Mo >= 32 && Mo <= 128 && (Mo & 31) == 0 && No >= 32 && No <= 128 && (No & 31) == 0 && P >= 32768 &&
           (P & 7) == 0 && P < (1ll << 30);   // row indices and pool offsets are 32-bit scalars in the kernel
}

long long prifit_gemm_stream_tn_workspace(int Mo, int No, long long P)
{
    long long per;
    return stream_tn_split(Mo, No, P, &per) * Mo * No;
}

int prifit_gemm_stream_tn_f32(int Mo, int No, long long P, const float *G, long long ldg, const float *A,
                              long long lda, float *out, long long ldo, const float *b_scale,
                              const float *b_shift, float *workspace, void *stream)
{
    if (!G || !A || !out || !workspace || !prifit_gemm_stream_tn_supported(Mo, No, P) || ldg < Mo || lda < No ||
        ldg >= (1 << 24) || lda >= (1 << 24) ||
        ldo < No || ((b_scale == nullptr) != (b_shift == nullptr)))
        return PRIFIT_EINVAL;
    StreamTNArgs g;
    g.G = G; g.A = A; g.ws = workspace; g.Mo = Mo; g.No = No; g.P = P; g.ldg = ldg; g.lda = lda;
    g.b_scale = b_scale; g.b_shift = b_shift;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = g.pool_d = nullptr; g.pool_K = 0;
    g.bn_G = nullptr; g.bn_s = g.bn_t = g.bn_a = g.bn_b = g.bn_d = nullptr;
    return stream_tn_launch(g, out, ldo, stream);
}

int prifit_gemm_stream_tn_pool_f32(int Mo, int No, long long P, const float *Y, long long ldy, const float *A,
                                   long long lda, float *out, long long ldo, const float *b_scale,
                                   const float *b_shift, const int32_t *pool_arg, const float *pool_T,
                                   const float *coef_b, const float *coef_d, int pool_K, float *workspace,
                                   void *stream)
{
    if (!Y || !A || !out || !workspace || !pool_arg || !pool_T || !coef_b || !coef_d ||
        !prifit_gemm_stream_tn_supported(Mo, No, P) || ldy < Mo || lda < No || ldo < No || ldy >= (1 << 24) || lda >= (1 << 24) ||
        ((b_scale == nullptr) != (b_shift == nullptr)) || pool_K < 8 || (pool_K & 7) || (P % pool_K))
        return PRIFIT_EINVAL;
    StreamTNArgs g;
    g.G = Y; g.A = A; g.ws = workspace; g.Mo = Mo; g.No = No; g.P = P; g.ldg = ldy; g.lda = lda;
    g.b_scale = b_scale; g.b_shift = b_shift;
    g.pool_arg = pool_arg; g.pool_T = pool_T; g.pool_b = coef_b; g.pool_d = coef_d; g.pool_K = pool_K;
    g.bn_G = nullptr; g.bn_s = g.bn_t = g.bn_a = g.bn_b = g.bn_d = nullptr;
    return stream_tn_launch(g, out, ldo, stream);
}

int prifit_gemm_stream_tn_bn_f32(int Mo, int No, long long P, const float *G, const float *Y, long long ldy, const float *A,
                                 long long lda, float *out, long long ldo, const float *b_scale, const float *b_shift,
                                 const float *scale, const float *shift, const float *coef_a, const float *coef_b,
                                 const float *coef_d, float *workspace, void *stream)
{
    if (!G || !Y || !A || !out || !workspace || !b_scale || !b_shift || !scale || !shift || !coef_a || !coef_b || !coef_d ||
        !prifit_gemm_stream_tn_supported(Mo, No, P) || ldy < Mo || lda < No || ldo < No || ldy >= (1 << 24) || lda >= (1 << 24))
        return PRIFIT_EINVAL;
    StreamTNArgs g;
    g.G = Y; g.A = A; g.ws = workspace; g.Mo = Mo; g.No = No; g.P = P; g.ldg = ldy; g.lda = lda;
    g.b_scale = b_scale; g.b_shift = b_shift;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = g.pool_d = nullptr; g.pool_K = 0;
    g.bn_G = G; g.bn_s = scale; g.bn_t = shift; g.bn_a = coef_a; g.bn_b = coef_b; g.bn_d = coef_d;
    return stream_tn_launch(g, out, ldo, stream);
}

}  // extern "C"
