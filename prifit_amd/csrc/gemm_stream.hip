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
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// WN: waves along N (2 or 4; 256 threads = 4 waves, WM = 4 / WN waves along M); KG = K / 8;
// BKC: B is [N][K] (NT) else [K][N] (NN); AFF: prologue on A
template <int WN, int KG, bool BKC, bool AFF, bool RED, bool POOL, bool PMAX>
__global__ __launch_bounds__(256, (KG <= 8 ? (RED ? 3 : 4) : (KG <= 12 ? (RED ? 2 : 3) : 2))) void gemm_stream_kernel(const StreamArgs g)
{
    constexpr int K = KG * 8;
    constexpr int WM = 4 / WN;
    constexpr int TM = SBM / (32 * WM);       // 32-row accumulator tiles per wave
    constexpr int LD = K + SPAD;
    constexpr int NV = SBM * K / 4 / 256;     // float4 staged per thread
    static_assert(NV * 256 * 4 == SBM * K, "tile divides evenly");
    __shared__ __attribute__((aligned(16))) float s_a[2][SBM * LD];
    __shared__ __attribute__((aligned(16))) float s_aff[2][K];   // POOL: [0] = b
    __shared__ __attribute__((aligned(16))) float s_pool[2][2][POOL ? K : 4];  // [stage][arg | T][channel]
    __shared__ float s_red[WM][2][32 * WN];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int col = 32 * wn + li;
    const bool col_ok = col < g.N;   // wave-uniform for N % 32 == 0

    if (AFF) {
        for (int t = threadIdx.x; t < K; t += 256) { s_aff[0][t] = g.a_scale[t]; s_aff[1][t] = g.a_shift[t]; }
    }
    if (POOL) {
        for (int t = threadIdx.x; t < K; t += 256) s_aff[0][t] = g.pool_b[t];
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
    float4 st_pool = make_float4(0.f, 0.f, 0.f, 0.f);
    // per-thread element offsets of its NV float4 inside a tile (32-bit: the launcher checks M * lda < 2^31)
    unsigned aoff[NV];
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int id = threadIdx.x + 256 * p;
        const int row = id / (K / 4), c4 = id - row * (K / 4);
        aoff[p] = (unsigned)row * (unsigned)g.lda + 4u * c4;
    }
    const unsigned last_row_off = (unsigned)(g.M - 1) * (unsigned)g.lda;
    auto load_tile = [&](int tile) {
        const int m0 = tile * SBM;
        if (POOL && threadIdx.x < 2 * (K / 4)) {  // the (arg, T) rows of this tile's pooling group: 2 x K values
            const int which = threadIdx.x / (K / 4), c4 = threadIdx.x - which * (K / 4);
            const float *src = which ? g.pool_T : reinterpret_cast<const float *>(g.pool_arg);
            st_pool = ld4(src + (long long)(m0 / g.pool_K) * K + 4 * c4);
        }
        const float *At = g.A + (long long)m0 * g.lda;   // wave-uniform tile base
        const bool full = m0 + SBM <= g.M;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            unsigned o = aoff[p];
            if (!full) {  // tail tile: rows beyond M re-read the last row (zeroed at the LDS store)
                const int id = threadIdx.x + 256 * p;
                const int row = id / (K / 4), c4 = id - row * (K / 4);
                if (m0 + row >= g.M) o = last_row_off - (unsigned)m0 * (unsigned)g.lda + 4u * c4;
            }
            st[p] = ld4(At + o);
        }
    };
    auto store_tile = [&](int tile, float *dst, int stage) {
        const int m0 = tile * SBM;
        if (POOL && threadIdx.x < 2 * (K / 4)) {
            const int which = threadIdx.x / (K / 4), c4 = threadIdx.x - which * (K / 4);
            *reinterpret_cast<float4 *>(&s_pool[stage][which][4 * c4]) = st_pool;
        }
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + 256 * p;
            const int row = id / (K / 4), c4 = id - row * (K / 4);
            float4 x = st[p];
            if (AFF) {
                const float4 s = *reinterpret_cast<const float4 *>(&s_aff[0][4 * c4]);
                const float4 t = *reinterpret_cast<const float4 *>(&s_aff[1][4 * c4]);
                x.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
                x.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
            }
            if (m0 + row >= g.M) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(dst + row * LD + 4 * c4) = x;
        }
    };

    int tile = blockIdx.x;
    if (tile < tiles) load_tile(tile);
    if (AFF || POOL) __syncthreads();  // s_aff visible before the first staging
    for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
        float *As = s_a[it & 1];
        store_tile(tile, As, it & 1);
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < tiles) load_tile(next);

        f32x16 acc[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        // RED: the previous layer's pre-activations under this wave's output elements, in flight during the MFMAs
        float yp[TM][16];
        if (RED && col_ok) {
            const int mr = tile * SBM + wm * 32 * TM + 4 * lh;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mr + 32 * a + (r & 3) + 8 * (r >> 2);
                    yp[a][r] = g.red_Y[(long long)(row < g.M ? row : g.M - 1) * g.ldry + col];
                }
        }
        const float *ap = As + (wm * 32 * TM + li) * LD + 4 * lh;
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            float4 fa[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const float4 *>(ap + a * 32 * LD + 8 * q);
            if (POOL) {
                const float4 b4 = *reinterpret_cast<const float4 *>(&s_aff[0][8 * q + 4 * lh]);
                const int4 w4 = *reinterpret_cast<const int4 *>(&s_pool[it & 1][0][8 * q + 4 * lh]);
                const float4 t4 = *reinterpret_cast<const float4 *>(&s_pool[it & 1][1][8 * q + 4 * lh]);
                const int kr0 = (tile * SBM) % g.pool_K + wm * 32 * TM + li;  // sample index of this lane's row in its group
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int kr = kr0 + 32 * a;
                    fa[a].x = fmaf(b4.x, fa[a].x, w4.x == kr ? t4.x : 0.f);
                    fa[a].y = fmaf(b4.y, fa[a].y, w4.y == kr ? t4.y : 0.f);
                    fa[a].z = fmaf(b4.z, fa[a].z, w4.z == kr ? t4.z : 0.f);
                    fa[a].w = fmaf(b4.w, fa[a].w, w4.w == kr ? t4.w : 0.f);
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, bf[q].x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, bf[q].y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, bf[q].z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, bf[q].w, acc[a], 0, 0, 0);
            }
        }
        // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        if (col_ok) {
            const int m0 = tile * SBM + wm * 32 * TM + 4 * lh;
            const bool full = tile * SBM + SBM <= g.M;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float vmax = -INFINITY, vmin = INFINITY;
                int imax = 0, imin = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * a + (r & 3) + 8 * (r >> 2);
                    if (!full && row >= g.M) continue;
                    const float v = acc[a][r] + bias;
                    csum += v;
                    csq += v * v;
                    g.C[(long long)row * g.ldc + col] = v;
                    if (PMAX) {  // rows ascend with r inside a lane: strict comparisons keep the first occurrence
                        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (v > vmax) { vmax = v; imax = ri; }
                        if (v < vmin) { vmin = v; imin = ri; }
                    }
                    if (RED) {
                        const float y = yp[a][r];
                        const float gm = fmaf(y, r_s, r_t) > 0.f ? v : 0.f;
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
    }
    if (RED) { csum = m1; csq = m2; }
    float *slab_out = RED ? g.red_slab : g.stats;
    if (slab_out) {
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (lh == 0) { s_red[wm][0][32 * wn + li] = csum; s_red[wm][1][32 * wn + li] = csq; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * g.N; t += 256) {
            const int which = t / g.N, c = t - which * g.N;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += s_red[w][which][c];
            slab_out[((long long)blockIdx.x * 2 + which) * g.N + c] = s;
        }
    }
}

// dW = G^T . relu(bn(A)) over the grouped samples: out[Mo, No] += sum_rows G[row, 0:Mo]^T A[row, 0:No]  (TN, the
// reduction runs over 10^5..10^6 rows, the output is at most 128 x 128).  Both operands stream; there is nothing to
// share between tiles, so there is no LDS and no barrier at all: every WAVE walks its own rows, loads the MFMA
// fragments straight from global memory (lane = output row / column, 2 x 128 contiguous bytes per load instruction),
// keeps one 8-row group in flight while the previous one is multiplied, and accumulates an output sub-block of at
// most 64 x 64 in registers.  The four waves of a workgroup cover (sub-blocks) x (interleaved row groups), so the
// second reader of a row finds it in L1 / L2.  The row-group streams of a workgroup are combined through LDS, every
// workgroup stores one partial [Mo, No] slab into the caller's workspace (plain stores: 768 workgroups hammering the
// same few KB with atomics cost more than the streaming itself), and a second tiny launch adds the slabs to `out`.
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
};

template <bool AFF, bool POOL>
__global__ __launch_bounds__(256, 3) void gemm_stream_tn_kernel(const StreamTNArgs g)
{
    __shared__ float s_part[3][64 * 64];  // partial sub-blocks of the row-group streams ks = 1..3
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int mt_total = g.Mo >> 5, nt_total = g.No >> 5;
    const int mb = (mt_total + 1) >> 1, nb = (nt_total + 1) >> 1;
    const int SB = mb * nb, KS = 4 / SB;          // sub-blocks (1, 2 or 4) x interleaved row-group streams
    const int sb = wave % SB, ks = wave / SB;
    const int mblk = sb / nb, nblk = sb - mblk * nb;
    const int m0 = 64 * mblk, n0 = 64 * nblk;
    const bool m2 = mt_total - 2 * mblk >= 2, n2 = nt_total - 2 * nblk >= 2;  // wave-uniform

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
    if (AFF) {
        sc[0] = g.b_scale[n0 + li]; sh[0] = g.b_shift[n0 + li];
        if (n2) { sc[1] = g.b_scale[n0 + 32 + li]; sh[1] = g.b_shift[n0 + 32 + li]; }
    }
    float pb[2] = {0.f, 0.f}, pd[2] = {0.f, 0.f};
    if (POOL) {
        pb[0] = g.pool_b[m0 + li]; pd[0] = g.pool_d[m0 + li];
        if (m2) { pb[1] = g.pool_b[m0 + 32 + li]; pd[1] = g.pool_d[m0 + 32 + li]; }
    }
    const long long r0 = (long long)blockIdx.x * g.rows_per_wg;
    const long long r1 = r0 + g.rows_per_wg < g.P ? r0 + g.rows_per_wg : g.P;
    const float *Gp = g.G + m0 + li + (long long)(4 * lh) * g.ldg;
    const float *Ap = g.A + n0 + li + (long long)(4 * lh) * g.lda;
    const long long stride = 8 * KS;

    float cg[2][4], ca[2][4], ng[2][4], na[2][4];
    int cw[2] = {0, 0}, nw[2] = {0, 0};          // POOL: winning sample of this lane's channel in the row group's pool
    float ct[2] = {0.f, 0.f}, nt[2] = {0.f, 0.f};  //       and its gradient
#pragma unroll
    for (int j = 0; j < 4; ++j) { cg[1][j] = 0.f; ca[1][j] = 0.f; ng[1][j] = 0.f; na[1][j] = 0.f; }
    auto load = [&](long long row, float (&fg)[2][4], float (&fa)[2][4], int (&fw)[2], float (&ft)[2]) {
        const float *gp = Gp + row * g.ldg;
        const float *ap = Ap + row * g.lda;
        if (POOL) {
            const long long o = (row / g.pool_K) * g.Mo + m0 + li;
            fw[0] = g.pool_arg[o]; ft[0] = g.pool_T[o];
            if (m2) { fw[1] = g.pool_arg[o + 32]; ft[1] = g.pool_T[o + 32]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fg[0][j] = gp[(long long)j * g.ldg];
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[0][j] = ap[(long long)j * g.lda];
        if (m2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fg[1][j] = gp[(long long)j * g.ldg + 32];
        }
        if (n2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[1][j] = ap[(long long)j * g.lda + 32];
        }
    };
    long long row = r0 + 8 * ks;
    if (row < r1) load(row, cg, ca, cw, ct);
    for (; row < r1; row += stride) {
        if (row + stride < r1) load(row + stride, ng, na, nw, nt);
        if (POOL) {
            const int kb = (int)(row % g.pool_K) + 4 * lh;  // sample index of this lane's first row in its pooling group
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cg[0][j] = fmaf(pb[0], cg[0][j], pd[0]) + (cw[0] == kb + j ? ct[0] : 0.f);
                cg[1][j] = fmaf(pb[1], cg[1][j], pd[1]) + (cw[1] == kb + j ? ct[1] : 0.f);
            }
        }
        if (AFF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ca[0][j] = fmaxf(fmaf(ca[0][j], sc[0], sh[0]), 0.f);
                ca[1][j] = fmaxf(fmaf(ca[1][j], sc[1], sh[1]), 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg[0][j], ca[0][j], acc[0][0], 0, 0, 0);
            if (n2) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg[0][j], ca[1][j], acc[0][1], 0, 0, 0);
            if (m2) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg[1][j], ca[0][j], acc[1][0], 0, 0, 0);
            if (m2 && n2) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cg[1][j], ca[1][j], acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { cg[0][j] = ng[0][j]; cg[1][j] = ng[1][j]; ca[0][j] = na[0][j]; ca[1][j] = na[1][j]; }
        if (POOL) { cw[0] = nw[0]; cw[1] = nw[1]; ct[0] = nt[0]; ct[1] = nt[1]; }
    }
    // combine the KS row-group streams of every sub-block through LDS (stream 0 of each sub-block collects)
    if (KS > 1) {
        if (ks > 0) {
            float *dst = s_part[wave - SB];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (ks > 0) return;
        for (int k = 1; k < KS; ++k) {
            const float *src = s_part[k * SB + sb - SB];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * 2 + b) * 16 + r) * 64 + lane];
        }
    }
    // C/D layout: col = lane & 31 (n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (m)
    float *slab = g.ws + (long long)blockIdx.x * g.Mo * g.No;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (a == 1 && !m2) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b == 1 && !n2) continue;
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
void launch_aff(const StreamArgs &g, int grid, hipStream_t st)
{
    if (!BKC && g.pool_arg && g.red_slab) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, true, true, false>), dim3(grid), dim3(256), 0, st, g);
    else if (!BKC && g.pool_arg) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, false, true, false>), dim3(grid), dim3(256), 0, st, g);
    else if (!BKC && g.red_slab) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, false, false, true, false, false>), dim3(grid), dim3(256), 0, st, g);
    else if (BKC && g.cand && g.a_scale) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, true, true, false, false, true>), dim3(grid), dim3(256), 0, st, g);
    else if (g.a_scale) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, BKC, true, false, false, false>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, BKC, false, false, false, false>), dim3(grid), dim3(256), 0, st, g);
}

template <int WN, bool BKC>
int launch_k(const StreamArgs &g, int grid, hipStream_t st)
{
    switch (g.K) {
        case 64: launch_aff<WN, 8, BKC>(g, grid, st); break;
        case 96: launch_aff<WN, 12, BKC>(g, grid, st); break;
        case 128: launch_aff<WN, 16, BKC>(g, grid, st); break;
        default: return PRIFIT_EINVAL;
    }
    return prifit_check_launch();
}

}  // namespace

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
    if ((long long)g.M * g.lda >= 2147483647LL) return PRIFIT_EINVAL;  // 32-bit tile offsets
    const int grid = stream_grid(g.M, g.K);
    hipStream_t st = as_stream(stream);
    if (g.N == 64) return layout == 0 ? launch_k<2, true>(g, grid, st) : launch_k<2, false>(g, grid, st);
    return layout == 0 ? launch_k<4, true>(g, grid, st) : launch_k<4, false>(g, grid, st);
}

int prifit_gemm_stream_dgrad_f32(int M, int N, int K, const float *dY, long long lda, const float *W, long long ldb,
                                 float *G, long long ldc, const float *Yprev, long long ldy, const float *scale,
                                 const float *shift, const float *mean, const float *invstd, float *red_slab,
                                 void *stream)
{
    if (!dY || !W || !G || !Yprev || !scale || !shift || !mean || !invstd || !red_slab ||
        !prifit_gemm_stream_supported(1, M, N, K) || lda < K || ldc < N || ldb < N || ldy < N || (lda & 3) ||
        ((uintptr_t)dY & 15))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.A = dY; g.B = W; g.C = G; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = nullptr; g.a_shift = nullptr; g.bias = nullptr; g.stats = nullptr;
    g.red_Y = Yprev; g.ldry = ldy; g.red_scale = scale; g.red_shift = shift; g.red_mean = mean; g.red_invstd = invstd;
    g.red_slab = red_slab;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = nullptr;
    return stream_launch(g, 1, stream);
}

int prifit_gemm_stream_dgrad_pool_f32(int M, int N, int K, const float *Y, long long lda, const float *W, long long ldb,
                                      float *G, long long ldc, const float *bias_dW, const int32_t *pool_arg,
                                      const float *pool_T, const float *coef_b, int pool_K, const float *Yprev,
                                      long long ldy, const float *scale, const float *shift, const float *mean,
                                      const float *invstd, float *red_slab, void *stream)
{
    if (!Y || !W || !G || !pool_arg || !pool_T || !coef_b || !prifit_gemm_stream_supported(1, M, N, K) || lda < K ||
        ldc < N || ldb < N || (lda & 3) || ((uintptr_t)Y & 15) || pool_K < 64 || (pool_K & 63) || (M % pool_K) ||
        (((uintptr_t)pool_arg | (uintptr_t)pool_T) & 15))
        return PRIFIT_EINVAL;
    if (red_slab && (!Yprev || !scale || !shift || !mean || !invstd || ldy < N)) return PRIFIT_EINVAL;
    StreamArgs g;
    g.A = Y; g.B = W; g.C = G; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = nullptr; g.a_shift = nullptr; g.bias = bias_dW; g.stats = nullptr;
    g.red_Y = Yprev; g.ldry = ldy; g.red_scale = scale; g.red_shift = shift; g.red_mean = mean; g.red_invstd = invstd;
    g.red_slab = red_slab;
    g.pool_arg = pool_arg; g.pool_T = pool_T; g.pool_b = coef_b; g.pool_K = pool_K; g.cand = nullptr;
    return stream_launch(g, 1, stream);
}

int prifit_gemm_stream_f32(int layout, int M, int N, int K, const float *A, long long lda, const float *B,
                           long long ldb, float *C, long long ldc, const float *a_scale, const float *a_shift,
                           const float *bias, float *col_stats, void *stream)
{
    if (!A || !B || !C || !prifit_gemm_stream_supported(layout, M, N, K) || lda < K || ldc < N ||
        ((a_scale == nullptr) != (a_shift == nullptr)) || (lda & 3) || ((uintptr_t)A & 15) ||
        (layout == 0 && ((ldb & 3) || ((uintptr_t)B & 15) || ldb < K)) || (layout == 1 && ldb < N))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = a_scale; g.a_shift = a_shift; g.bias = bias; g.stats = col_stats;
    g.red_Y = nullptr; g.ldry = 0; g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr; g.red_slab = nullptr;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = nullptr;
    return stream_launch(g, layout, stream);
}

int prifit_gemm_stream_pool_f32(int M, int N, int K, const float *A, long long lda, const float *B, long long ldb,
                                float *C, long long ldc, const float *a_scale, const float *a_shift, const float *bias,
                                float *col_stats, float *cand, void *stream)
{
    if (!A || !B || !C || !cand || !a_scale || !a_shift || !prifit_gemm_stream_supported(0, M, N, K) || lda < K ||
        ldc < N || (lda & 3) || ((uintptr_t)A & 15) || (ldb & 3) || ((uintptr_t)B & 15) || ldb < K || (M & 31))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = a_scale; g.a_shift = a_shift; g.bias = bias; g.stats = col_stats;
    g.red_Y = nullptr; g.ldry = 0; g.red_scale = g.red_shift = g.red_mean = g.red_invstd = nullptr; g.red_slab = nullptr;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = nullptr; g.pool_K = 0; g.cand = cand;
    return stream_launch(g, 0, stream);
}

int prifit_gemm_stream_tn_supported(int Mo, int No, long long P)
{
    return Mo >= 32 && Mo <= 128 && (Mo & 31) == 0 && No >= 32 && No <= 128 && (No & 31) == 0 && P >= 32768 &&
           (P & 7) == 0;
}

static long long stream_tn_split(long long P, long long *per_out)
{
    long long nwg = 768;                        // 3 resident workgroups per CU (about 140 VGPRs per wave)
    long long per = (P + nwg - 1) / nwg;
    per = (per + 31) / 32 * 32;                 // whole 8-row groups for every interleaved stream
    *per_out = per;
    return (P + per - 1) / per;
}

long long prifit_gemm_stream_tn_workspace(int Mo, int No, long long P)
{
    long long per;
    return stream_tn_split(P, &per) * Mo * No;
}

static int stream_tn_launch(StreamTNArgs &g, float *out, long long ldo, void *stream)
{
    const long long nwg = stream_tn_split(g.P, &g.rows_per_wg);
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)nwg), block(256);
    if (g.pool_arg) {
        if (g.b_scale) hipLaunchKernelGGL((gemm_stream_tn_kernel<true, true>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_stream_tn_kernel<false, true>), grid, block, 0, st, g);
    } else {
        if (g.b_scale) hipLaunchKernelGGL((gemm_stream_tn_kernel<true, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm_stream_tn_kernel<false, false>), grid, block, 0, st, g);
    }
    hipLaunchKernelGGL(stream_tn_reduce_kernel, dim3((g.Mo * g.No + 255) / 256, 32), dim3(256), 0, st, g.ws, (int)nwg,
                       g.Mo, g.No, ldo, out);
    return prifit_check_launch();
}

int prifit_gemm_stream_tn_f32(int Mo, int No, long long P, const float *G, long long ldg, const float *A,
                              long long lda, float *out, long long ldo, const float *b_scale,
                              const float *b_shift, float *workspace, void *stream)
{
    if (!G || !A || !out || !workspace || !prifit_gemm_stream_tn_supported(Mo, No, P) || ldg < Mo || lda < No ||
        ldo < No || ((b_scale == nullptr) != (b_shift == nullptr)))
        return PRIFIT_EINVAL;
    StreamTNArgs g;
    g.G = G; g.A = A; g.ws = workspace; g.Mo = Mo; g.No = No; g.P = P; g.ldg = ldg; g.lda = lda;
    g.b_scale = b_scale; g.b_shift = b_shift;
    g.pool_arg = nullptr; g.pool_T = nullptr; g.pool_b = g.pool_d = nullptr; g.pool_K = 0;
    return stream_tn_launch(g, out, ldo, stream);
}

int prifit_gemm_stream_tn_pool_f32(int Mo, int No, long long P, const float *Y, long long ldy, const float *A,
                                   long long lda, float *out, long long ldo, const float *b_scale,
                                   const float *b_shift, const int32_t *pool_arg, const float *pool_T,
                                   const float *coef_b, const float *coef_d, int pool_K, float *workspace,
                                   void *stream)
{
    if (!Y || !A || !out || !workspace || !pool_arg || !pool_T || !coef_b || !coef_d ||
        !prifit_gemm_stream_tn_supported(Mo, No, P) || ldy < Mo || lda < No || ldo < No ||
        ((b_scale == nullptr) != (b_shift == nullptr)) || pool_K < 8 || (pool_K & 7) || (P % pool_K))
        return PRIFIT_EINVAL;
    StreamTNArgs g;
    g.G = Y; g.A = A; g.ws = workspace; g.Mo = Mo; g.No = No; g.P = P; g.ldg = ldy; g.lda = lda;
    g.b_scale = b_scale; g.b_shift = b_shift;
    g.pool_arg = pool_arg; g.pool_T = pool_T; g.pool_b = coef_b; g.pool_d = coef_d; g.pool_K = pool_K;
    return stream_tn_launch(g, out, ldo, stream);
}

}  // extern "C"
