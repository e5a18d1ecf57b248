// Row-sparse backward of the mean-shift iterations (src/mean_shift.py:44-46, :61-82).
//
// What the loss reads of the shifted points is `center = new_X[indices]` (:46) -- at most max_num_clusters rows per
// shape -- and row i of an iterate depends on row i of the previous iterate alone: the dictionary is the fixed input X
// (`new_X @ X^T`, :65), not the moving points.  d loss / d new_X is therefore non-zero on the kept rows ONLY, in every
// iteration, and the dense backward (four N x N x D products per iteration, of which N - R query rows multiply exact
// zeros) collapses to R x N x D products, R <= 32 of N = 2048.  The result is the dense one: the rows that are left out
// contribute exact zeros there.  The forward then has no reason to keep K^T (403 MB per iteration at B = 24): the K
// values under the kept rows are re-formed here from the saved iterate rows.
//
// Per iteration t (last to first), R rows per shape:
//   rows_prep:  g_r = dL/d(row r of iterate t + 1)   (the caller's gradient for the last iteration, else the sum of the
//               per-key-tile partials the main kernel of iteration t + 1 wrote);  through the update
//               out = normalize(Z + (O / rowsum - Z)) (:70-82) to gO_r [D] and g_rowsum_r;  z_r = row of iterate t.
//   rows_main:  one workgroup per tile of 64 keys j:  s_rj = z_r . x_j,  K_rj = exp(clamp((s_rj - 1) / b^2, -13, 75)),
//               gS_rj = (gO_r . x_j + g_rowsum_r) K_rj / b^2 where the lower clamp is inactive (src/guard.py:6-11),
//               dX_j += sum_r gS_rj z_r + K_rj gO_r   (the tile owns its keys: plain read-modify-write, iterations are
//               consecutive launches),   partial dZ_r = sum_{j in tile} gS_rj x_j  (one slab per tile, summed by the next
//               rows_prep: deterministic, no atomics).
//   rows_final: dX[id_r] += dL/d(row r of iterate 0)   (Z_0 = X.clone(), :60).
// fp32 VALU throughout: R is the small dimension (1 .. 32, a run-time count per shape), the products are R/N of a
// dense iteration and a 32-row MFMA tile would spend the matrix pipe on padding.
#include "common.h"

namespace {

constexpr int KT = 64;        // keys per workgroup
constexpr int RC = 8;         // rows per chunk of the score phase
constexpr float LOG2E = 1.44269504088896341f;

struct RowsArgs {
    const float *X, *bw;
    const float *Zin, *Zout, *O, *rsum, *nrm;   // this iteration's saved tensors
    const long long *ids;
    const int *nrows;
    const float *g_rows;                        // first (= last iteration) launch: the caller's gradient, else NULL
    float *gO, *grs, *zrow, *part, *dX;
    int B, N, R, ntile;
};

__device__ __forceinline__ int live_rows(const RowsArgs &a, int b)
{
    const int n = a.nrows ? a.nrows[b] : a.R;
    return n < 0 ? 0 : (n > a.R ? a.R : n);
}

// grid (R, B), 256 threads per (row, shape): thread = (column c, tile lane q of 256 / D): the per-tile partials are summed by
// 256 / D lanes per column and combined through LDS in a fixed order (deterministic), then one wave finishes the row
template <int D>
__global__ __launch_bounds__(256) void ms_rows_prep_kernel(RowsArgs a)
{
    constexpr int TL = 256 / D;                   // tile lanes per column (2 at D = 128, 8 at D = 32)
    constexpr int PL = D / 64 > 0 ? D / 64 : 1;   // columns per lane of the finishing wave
    __shared__ float s_g[TL][D];
    const int r = blockIdx.x, b = blockIdx.y;
    if (r >= live_rows(a, b)) return;
    long long id = a.ids[(size_t)b * a.R + r];
    id = id < 0 ? 0 : (id >= a.N ? a.N - 1 : id);
    const size_t row = (size_t)b * a.N + id;
    const size_t slot = (size_t)b * a.R + r;
    {
        const int c = threadIdx.x % D, q = threadIdx.x / D;
        float v = 0.f;
        if (a.g_rows) v = q == 0 ? a.g_rows[slot * D + c] : 0.f;
        else
            for (int t = q; t < a.ntile; t += TL) v += a.part[(((size_t)b * a.ntile + t) * a.R + r) * D + c];
        s_g[q][c] = v;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    float g[PL], o[PL];
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
        const int c = lane + 64 * p;
        float v = 0.f;
        if (c < D)
#pragma unroll
            for (int q = 0; q < TL; ++q) v += s_g[q][c];
        g[p] = v;
        o[p] = c < D ? a.Zout[row * D + c] : 0.f;
        dot += g[p] * o[p];
    }
    dot = wave_sum_f32(dot);
    const float rinv = 1.0f / a.rsum[row], ninv = 1.0f / a.nrm[row];
    float gr = 0.f;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
        const int c = lane + 64 * p;
        if (c < D) {
            const float gnew = (g[p] - o[p] * dot) * ninv;       // through the normalisation
            a.gO[slot * D + c] = gnew * rinv;                   // d/dO of O / rowsum
            gr -= gnew * (a.O[row * D + c] * rinv);
            a.zrow[slot * D + c] = a.Zin[row * D + c];
        }
    }
    gr = wave_sum_f32(gr);
    if (lane == 0) a.grs[slot] = gr * rinv;                     // d/d(rowsum); d/dZ through "Z + (Mv - Z)" is exactly 0
}

// grid (R, B): dX[id_r] += sum of the last main launch's partials
template <int D>
__global__ __launch_bounds__(64) void ms_rows_final_kernel(RowsArgs a)
{
    const int r = blockIdx.x, b = blockIdx.y;
    if (r >= live_rows(a, b)) return;
    long long id = a.ids[(size_t)b * a.R + r];
    id = id < 0 ? 0 : (id >= a.N ? a.N - 1 : id);
    for (int c = threadIdx.x; c < D; c += 64) {
        float v = 0.f;
        if (a.g_rows) v = a.g_rows[((size_t)b * a.R + r) * D + c];   // zero iterations: the gather itself
        else
            for (int t = 0; t < a.ntile; ++t) v += a.part[(((size_t)b * a.ntile + t) * a.R + r) * D + c];
        unsafeAtomicAdd(a.dX + ((size_t)b * a.N + id) * D + c, v);   // (two slots may name the same point)
    }
}

// grid (ntile, B), 256 threads
template <int D>
__global__ __launch_bounds__(256) void ms_rows_main_kernel(RowsArgs a)
{
    constexpr int LDX = D + 4;             // padded key rows: conflict-free float4 reads down a column of keys
    constexpr int DQ = D / 4;              // score phase: columns per wave
    constexpr int ND4 = D / 4;             // float4 per row
    constexpr int NG = 256 / ND4;          // thread groups over a row's float4s (8 at D = 128)
    constexpr int KPG = KT / NG;           // keys per group in the dX phase
    constexpr int RMAX = 32;
    __shared__ __attribute__((aligned(16))) float s_x[KT * LDX];
    __shared__ float s_p[2][4][RC][KT];    // partial scores (s, t) per column quarter
    __shared__ __attribute__((aligned(16))) float s_gs[RMAX][KT], s_k[RMAX][KT];

    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nr = live_rows(a, b);
    if (nr == 0) return;                   // uniform: no gradient enters this shape's trajectory
    const int k0 = tile * KT;
    const float *Xb = a.X + (size_t)b * a.N * D;
    const float bwv = a.bw[b];
    const float rcp_b2 = 1.0f / (bwv * bwv), c_e2 = rcp_b2 * LOG2E;
    const float kmin = __expf(-13.0f);

    // ---- key tile -> LDS (rows beyond N as zeros)
    for (int i = tid; i < KT * ND4; i += 256) {
        const int j = i / ND4, c4 = i - j * ND4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + j < a.N) v = *reinterpret_cast<const float4 *>(Xb + (size_t)(k0 + j) * D + 4 * c4);
        *reinterpret_cast<float4 *>(s_x + j * LDX + 4 * c4) = v;
    }
    __syncthreads();

    // ---- scores: thread (key j, column quarter dq = wave); the rows' operands are wave-uniform (scalar loads)
    const int j = tid & 63, dq = __builtin_amdgcn_readfirstlane(tid >> 6);
    float xr[DQ];
#pragma unroll
    for (int i = 0; i < DQ; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(s_x + j * LDX + dq * DQ + i);
        xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
    }
    const float *zr_base = a.zrow + (size_t)b * a.R * D + dq * DQ;
    const float *go_base = a.gO + (size_t)b * a.R * D + dq * DQ;
    for (int r0 = 0; r0 < nr; r0 += RC) {
        const int rn = min(RC, nr - r0);
        for (int rr = 0; rr < rn; ++rr) {
            const float *zr = zr_base + (size_t)(r0 + rr) * D, *go = go_base + (size_t)(r0 + rr) * D;
            float s = 0.f, t = 0.f;
#pragma unroll
            for (int i = 0; i < DQ; ++i) {
                s = fmaf(zr[i], xr[i], s);
                t = fmaf(go[i], xr[i], t);
            }
            s_p[0][dq][rr][j] = s;
            s_p[1][dq][rr][j] = t;
        }
        __syncthreads();
        for (int e = tid; e < rn * KT; e += 256) {
            const int rr = e >> 6, jj = e & 63, r = r0 + rr;
            const float s = (s_p[0][0][rr][jj] + s_p[0][1][rr][jj]) + (s_p[0][2][rr][jj] + s_p[0][3][rr][jj]);
            const float t = (s_p[1][0][rr][jj] + s_p[1][1][rr][jj]) + (s_p[1][2][rr][jj] + s_p[1][3][rr][jj]);
            // the forward's transform (meanshift_fused.hip): exp2(clamp((s - 1) log2(e) / b^2, -13 log2(e), 75 log2(e)))
            const float u = fminf(fmaxf(fmaf(s, c_e2, -c_e2), -13.0f * LOG2E), 75.0f * LOG2E);
            const float kv = (k0 + jj < a.N) ? __builtin_amdgcn_exp2f(u) : 0.f;
            s_k[r][jj] = kv;
            s_gs[r][jj] = kv > kmin ? (t + a.grs[(size_t)b * a.R + r]) * kv * rcp_b2 : 0.f;
        }
        __syncthreads();
    }

    // ---- partial dZ_r = sum_j gS_rj x_j: thread (float4 column c4, row group rq); rows rq, rq + NG, ...
    const int c4 = tid % ND4, grp = tid / ND4;
    constexpr int RPT = RMAX / NG > 0 ? RMAX / NG : 1;   // rows per thread (4 at D = 128)
    {
        float4 acc[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int jj = 0; jj < KT; jj += 4) {
            float4 xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const float4 *>(s_x + (jj + q) * LDX + 4 * c4);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = grp + NG * i;
                if (r < nr) {
                    const float4 gs = *reinterpret_cast<const float4 *>(&s_gs[r][jj]);
                    acc[i].x = fmaf(gs.x, xv[0].x, acc[i].x); acc[i].y = fmaf(gs.x, xv[0].y, acc[i].y);
                    acc[i].z = fmaf(gs.x, xv[0].z, acc[i].z); acc[i].w = fmaf(gs.x, xv[0].w, acc[i].w);
                    acc[i].x = fmaf(gs.y, xv[1].x, acc[i].x); acc[i].y = fmaf(gs.y, xv[1].y, acc[i].y);
                    acc[i].z = fmaf(gs.y, xv[1].z, acc[i].z); acc[i].w = fmaf(gs.y, xv[1].w, acc[i].w);
                    acc[i].x = fmaf(gs.z, xv[2].x, acc[i].x); acc[i].y = fmaf(gs.z, xv[2].y, acc[i].y);
                    acc[i].z = fmaf(gs.z, xv[2].z, acc[i].z); acc[i].w = fmaf(gs.z, xv[2].w, acc[i].w);
                    acc[i].x = fmaf(gs.w, xv[3].x, acc[i].x); acc[i].y = fmaf(gs.w, xv[3].y, acc[i].y);
                    acc[i].z = fmaf(gs.w, xv[3].z, acc[i].z); acc[i].w = fmaf(gs.w, xv[3].w, acc[i].w);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = grp + NG * i;
            if (r < nr)
                *reinterpret_cast<float4 *>(a.part + (((size_t)b * a.ntile + tile) * a.R + r) * D + 4 * c4) = acc[i];
        }
    }

    // ---- dX_j += sum_r gS_rj z_r + K_rj gO_r: thread (float4 column c4, key group grp: keys grp * KPG ..)
    {
        float4 acc[KPG];
#pragma unroll
        for (int q = 0; q < KPG; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *zr = a.zrow + (size_t)b * a.R * D + 4 * c4, *go = a.gO + (size_t)b * a.R * D + 4 * c4;
        for (int r = 0; r < nr; ++r) {
            const float4 z = *reinterpret_cast<const float4 *>(zr + (size_t)r * D);
            const float4 g = *reinterpret_cast<const float4 *>(go + (size_t)r * D);
#pragma unroll
            for (int q = 0; q < KPG; ++q) {
                const float gs = s_gs[r][grp * KPG + q], kv = s_k[r][grp * KPG + q];
                acc[q].x = fmaf(gs, z.x, fmaf(kv, g.x, acc[q].x));
                acc[q].y = fmaf(gs, z.y, fmaf(kv, g.y, acc[q].y));
                acc[q].z = fmaf(gs, z.z, fmaf(kv, g.z, acc[q].z));
                acc[q].w = fmaf(gs, z.w, fmaf(kv, g.w, acc[q].w));
            }
        }
#pragma unroll
        for (int q = 0; q < KPG; ++q) {
            const int key = k0 + grp * KPG + q;
            if (key < a.N) {
                float4 *p = reinterpret_cast<float4 *>(a.dX + ((size_t)b * a.N + key) * D + 4 * c4);
                float4 v = *p;
                v.x += acc[q].x; v.y += acc[q].y; v.z += acc[q].z; v.w += acc[q].w;
                *p = v;
            }
        }
    }
}

template <int D>
int rows_bwd(const float *X, const float *bw, int B, int N, int T, const float *const *Zin, const float *const *Zout,
             const float *const *O, const float *const *rsum, const float *const *nrm, const long long *ids,
             const int *nrows, int R, const float *g_rows, float *ws, float *dX, hipStream_t st)
{
    RowsArgs a;
    a.X = X; a.bw = bw; a.ids = ids; a.nrows = nrows; a.dX = dX; a.B = B; a.N = N; a.R = R;
    a.ntile = (N + KT - 1) / KT;
    const size_t rd = (size_t)B * R * D;
    a.gO = ws; a.zrow = ws + rd; a.grs = ws + 2 * rd; a.part = ws + 2 * rd + (((size_t)B * R + 3) & ~(size_t)3);
    a.g_rows = g_rows;
    a.Zin = a.Zout = a.O = a.rsum = a.nrm = nullptr;
    for (int t = T - 1; t >= 0; --t) {
        a.Zin = Zin[t]; a.Zout = Zout[t]; a.O = O[t]; a.rsum = rsum[t]; a.nrm = nrm[t];
        if (!a.Zin || !a.Zout || !a.O || !a.rsum || !a.nrm) return PRIFIT_EINVAL;
        hipLaunchKernelGGL(ms_rows_prep_kernel<D>, dim3(R, B), dim3(256), 0, st, a);
        hipLaunchKernelGGL(ms_rows_main_kernel<D>, dim3(a.ntile, B), dim3(256), 0, st, a);
        a.g_rows = nullptr;
    }
    hipLaunchKernelGGL(ms_rows_final_kernel<D>, dim3(R, B), dim3(64), 0, st, a);
    return prifit_check_launch();
}

}  // namespace

extern "C" {

int prifit_meanshift_rows_supported(int N, int D, int R)
{
    return (N > 0 && R >= 1 && R <= 32 && (D == 32 || D == 64 || D == 128)) ? 1 : 0;
}

long long prifit_meanshift_rows_bwd_workspace(int B, int N, int D, int R)
{
    if (B <= 0 || !prifit_meanshift_rows_supported(N, D, R)) return 0;
    const long long ntile = (N + KT - 1) / KT, rd = (long long)B * R * D;
    return 2 * rd + (((long long)B * R + 3) & ~3LL) + (long long)B * ntile * R * D;
}

int prifit_meanshift_rows_bwd(const float *X, const float *bw, int B, int N, int D, int T, const float *const *Zin,
                              const float *const *Zout, const float *const *O, const float *const *rowsum,
                              const float *const *nrm, const long long *ids, const int *nrows, int R,
                              const float *g_rows, float *workspace, float *dX, void *stream)
{
    if (!X || !bw || !ids || !g_rows || !workspace || !dX || B <= 0 || B > 65535 || T < 0 || (T > 0 && (!Zin || !Zout || !O || !rowsum || !nrm)) ||
        !prifit_meanshift_rows_supported(N, D, R) || (((uintptr_t)X | (uintptr_t)dX | (uintptr_t)workspace) & 15))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    switch (D) {
    case 128: return rows_bwd<128>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, st);
    case 64: return rows_bwd<64>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, st);
    default: return rows_bwd<32>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, st);
    }
}

}  // extern "C"
