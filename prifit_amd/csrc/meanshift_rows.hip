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
// Launches (round 5: T + 2 of them; before: 2 T + 1, with dX read-modified-written by every iteration):
//   rows_first: per shape, g_r = the caller's gradient dL/d(row r of iterate T) through the update
//               out = normalize(Z + (O / rowsum - Z)) (:70-82) of the LAST iteration -> gO_r [D], g_rowsum_r, z_r = the row of
//               iterate T - 1 (`prep_rows`).
//   rows_iter (iteration t = T - 1 .. 0), one workgroup per tile of 64 keys j:
//               s_rj = z_r . x_j,  K_rj = exp(clamp((s_rj - 1) / b^2, -13, 75)),
//               gS_rj = (gO_r . x_j + g_rowsum_r) K_rj / b^2 where the lower clamp is inactive (src/guard.py:6-11);
//               K and gS go to a table [t][shape][row][key] -- dX is NOT touched here;
//               partial dZ_r = sum_{j in tile} gS_rj x_j, one slab per tile;
//               the workgroup that finishes LAST in its shape (a ticket per shape: fence, atomic, fence -- nobody waits) sums
//               the slabs in tile order and runs `prep_rows` for iteration t - 1 (t = 0: keeps dL/d(row of iterate 0));
//               deterministic: which workgroup does it changes nothing of what it computes.
//   rows_apply: per key tile, once:  dX_j += sum_t sum_r gS^t_rj z^t_r + K^t_rj gO^t_r  (+ dL/d(row of iterate 0) on the
//               kept points: Z_0 = X.clone(), :60), accumulators in registers, dX read-modified-written ONCE.
// fp32 VALU throughout: R is the small dimension (1 .. 32, a run-time count per shape), the products are R/N of a
// dense iteration and a 32-row MFMA tile would spend the matrix pipe on padding.
#include "common.h"

namespace {

constexpr int KT = 64;        // keys per workgroup
#ifndef MSR_RC
#define MSR_RC 4
#endif
constexpr int RC = MSR_RC;    // rows per chunk of the score phase (LDS: 50 KB per workgroup, three per CU: the 768 of B = 24 in one round)
constexpr int RCAP = 64;      // live rows per shape: kernels are instantiated for RMAX = 32 (the loss path: <= 25 clusters, KM = 32 slots)
                              // and RMAX = 64 (max_num_clusters up to 64: the reference's own guard uses 49, src/mean_shift.py:212-226)
constexpr int CH = 8;         // rows per ticket: the slabs of a shape are summed per chunk of 8 rows, by whichever workgroup finishes it last
constexpr int NCH_CAP = RCAP / CH;
constexpr float LOG2E = 1.44269504088896341f;
#ifndef MSR_PROBE
#define MSR_PROBE 0           // timing probes only (wrong results): 1 no ticket / tail, 2 also no partial dZ, 3 the key tile load alone
#endif

struct RowsArgs {
    const float *X, *bw;
    const long long *ids;
    const int *nrows;
    int B, N, R, ntile, T;
    // rows_iter, iteration t: where its partials go, the shape tickets (the prepared rows [B][R][D] x 2, [B][R] and the tables
    // [B][R][N] x 2 it writes are kernel parameters of their own)
    float *part;                                        // [B][ntile][R][D]
    int *counter;                                       // [B][NCH_CAP]
    // what prep_rows reads and writes: the saved tensors of the iteration it prepares (all NULL: iterate 0 is reached)
    const float *pZin, *pZout, *pO, *prsum, *pnrm;
    float *p_zrow, *p_gO, *p_grs;
    float *g0;                                          // [B][R][D]: dL/d(kept rows of iterate 0)
    const float *g_rows;                                // rows_first: the caller's gradient
    // rows_apply: the T tables
    const float *all_zrow, *all_gO, *all_gs, *all_k;    // [T][B][R][D] x 2, [T][B][R][N] x 2
    float *dX;
};

__device__ __forceinline__ int live_rows(const RowsArgs &a, int b)
{
    const int n = a.nrows ? a.nrows[b] : a.R;
    return n < 0 ? 0 : (n > a.R ? a.R : n);
}

__device__ __forceinline__ long long row_id(const RowsArgs &a, int b, int r)
{
    const long long id = a.ids[(size_t)b * a.R + r];
    return id < 0 ? 0 : (id >= a.N ? a.N - 1 : id);
}

// The slabs cross workgroups (and the eight L2s of the chip) INSIDE a launch: written and read as agent-scope accesses, which go
// past the per-XCD L2 -- a release / acquire fence pair would do, but on this chip a fence writes back and invalidates the
// whole L2 of its XCD: measured 77 us per iteration launch with `__threadfence()` in 768 workgroups, the kernel's own work ~10.
__device__ __forceinline__ float ld_agent(const float *p)
{
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<int *>(const_cast<float *>(p)), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ float2 ld_agent2(const float *p)                // p 8-byte aligned
{
    const unsigned long long bits = __hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<float *>(p)),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (unsigned)bits), __builtin_bit_cast(float, (unsigned)(bits >> 32)));
}

__device__ __forceinline__ void st_agent2(float *p, float x, float y)      // p 8-byte aligned
{
    const unsigned long long bits = ((unsigned long long)__builtin_bit_cast(unsigned, y) << 32) | __builtin_bit_cast(unsigned, x);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup of 256 threads, shape b, live rows [r0, r1): wave w takes the rows r0 + w, r0 + w + 4, ...; lane = a PAIR of
// columns (D <= 128: one 8-byte access per lane and row).  g_r = the caller's gradient (from_rows) or the sum of the ntile partial
// slabs -- all of them in flight at once, an agent-scope load being a ~2 us round trip past the L2; chain q sums the tiles q, q + 4,
// ... in ascending order, whoever runs this -- then through the update of the prepared iteration to (gO_r, g_rowsum_r), z_r = the
// row of that iteration's input.
template <int D>
__device__ __forceinline__ void prep_rows(const RowsArgs &a, int b, int r0, int r1, const float *from_rows)
{
    static_assert(D <= 128 && D % 2 == 0, "one column pair per lane");
    constexpr int SB = 32;                                   // slabs in flight
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = 2 * lane;
    const bool on = c < D;
    for (int r = r0 + wave; r < r1; r += 4) {
        const size_t slot = (size_t)b * a.R + r;
        // the prepared iteration's own rows first: their latency runs under the slab sums
        const size_t row = (size_t)b * a.N + row_id(a, b, r);
        float2 o = make_float2(0.f, 0.f), oo = o, zi = o;
        float rs = 1.f, nm = 1.f;
        if (a.pZin) {
            if (on) {
                o = *reinterpret_cast<const float2 *>(a.pZout + row * D + c);
                oo = *reinterpret_cast<const float2 *>(a.pO + row * D + c);
                zi = *reinterpret_cast<const float2 *>(a.pZin + row * D + c);
            }
            rs = a.prsum[row]; nm = a.pnrm[row];
        }
        float2 g = make_float2(0.f, 0.f);
        if (on) {
            if (from_rows) g = *reinterpret_cast<const float2 *>(from_rows + slot * D + c);
            else {
                const float *src = a.part + ((size_t)b * a.ntile * a.R + r) * D + c;
                const size_t st = (size_t)a.R * D;
                float2 ch[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ch[q] = make_float2(0.f, 0.f);
                for (int t0 = 0; t0 < a.ntile; t0 += SB) {
                    float2 w[SB];
#pragma unroll
                    for (int i = 0; i < SB; ++i) w[i] = t0 + i < a.ntile ? ld_agent2(src + (size_t)(t0 + i) * st) : make_float2(0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < SB; ++i) { ch[i & 3].x += w[i].x; ch[i & 3].y += w[i].y; }
                }
                g.x = (ch[0].x + ch[1].x) + (ch[2].x + ch[3].x);
                g.y = (ch[0].y + ch[1].y) + (ch[2].y + ch[3].y);
            }
        }
        if (!a.pZin) {                                           // iterate 0 = the input rows themselves
            if (on) *reinterpret_cast<float2 *>(a.g0 + slot * D + c) = g;
            continue;
        }
        const float dot = wave_sum_f32(g.x * o.x + g.y * o.y);
        const float rinv = 1.0f / rs, ninv = 1.0f / nm;
        float gr = 0.f;
        if (on) {
            const float gx = (g.x - o.x * dot) * ninv, gy = (g.y - o.y * dot) * ninv;      // through the normalisation
            *reinterpret_cast<float2 *>(a.p_gO + slot * D + c) = make_float2(gx * rinv, gy * rinv);   // d/dO of O / rowsum
            gr = -(gx * (oo.x * rinv) + gy * (oo.y * rinv));
            *reinterpret_cast<float2 *>(a.p_zrow + slot * D + c) = zi;
        }
        gr = wave_sum_f32(gr);
        if (lane == 0) a.p_grs[slot] = gr * rinv;                   // d/d(rowsum); d/dZ through "Z + (Mv - Z)" is exactly 0
    }
}

// grid (B), 256 threads: the caller's gradient through the last iteration (or, with no iteration, straight to g0); the
// shape's ticket starts at zero
template <int D>
__global__ __launch_bounds__(256) void ms_rows_first_kernel(RowsArgs a)
{
    const int b = blockIdx.x;
    if (threadIdx.x < NCH_CAP) a.counter[b * NCH_CAP + threadIdx.x] = 0;
    prep_rows<D>(a, b, 0, live_rows(a, b), a.g_rows);
}

// grid (ntile, B), 256 threads: one iteration
template <int D, int RMAX>
__global__ __launch_bounds__(256) void ms_rows_iter_kernel(RowsArgs a, const float *__restrict__ zrow, const float *__restrict__ gO,
                                                           const float *__restrict__ grs, float *__restrict__ coef_gs,
                                                           float *__restrict__ coef_k)
{
    constexpr int LDX = D + 4;             // padded key rows: conflict-free float4 reads down a column of keys
    constexpr int DQ = D / 4;              // score phase: columns per wave
    constexpr int ND4 = D / 4;             // float4 per row
    constexpr int NG = 256 / ND4;          // thread groups over a row's float4s (8 at D = 128)
    __shared__ __attribute__((aligned(16))) float s_x[KT * LDX];
    __shared__ float s_p[2][4][RC][KT];    // partial scores (s, t) per column quarter
    __shared__ __attribute__((aligned(16))) float s_gs[RMAX][KT];
    __shared__ int s_last;

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

    if (MSR_PROBE == 3) return;
    // ---- scores: thread (key j, column quarter dq = wave); the rows' operands are wave-uniform (scalar loads)
    const int j = tid & 63, dq = __builtin_amdgcn_readfirstlane(tid >> 6);
    float xr[DQ];
#pragma unroll
    for (int i = 0; i < DQ; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(s_x + j * LDX + dq * DQ + i);
        xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
    }
    const float *zr_base = zrow + (size_t)b * a.R * D + dq * DQ;
    const float *go_base = gO + (size_t)b * a.R * D + dq * DQ;
    // (the operands and the tables are `__restrict__` kernel parameters of their own, not members of `a`: with a global store in
    // this loop that the compiler cannot tell apart from them, the rows' scalar loads become per-lane vector loads of one address
    // -- measured 49 instead of ~30 us per launch at 25 rows)
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
            const float gs = kv > kmin ? (t + grs[(size_t)b * a.R + r]) * kv * rcp_b2 : 0.f;
            s_gs[r][jj] = gs;
            if (k0 + jj < a.N) {
                const size_t at = ((size_t)b * a.R + r) * a.N + k0 + jj;
                coef_gs[at] = gs;
                coef_k[at] = kv;
            }
        }
        __syncthreads();
    }

    if (MSR_PROBE == 2) return;
    // ---- partial dZ_r = sum_j gS_rj x_j: thread (float4 column c4, row group grp); rows grp, grp + NG, ... -- at D = 128 (NG = CH = 8)
    // ONE row of every chunk of CH rows, all of them in one pass over the key tile
    const int c4 = tid % ND4, grp = tid / ND4;
    constexpr int RPT = RMAX / NG > 0 ? RMAX / NG : 1;   // rows per thread (4 at D = 128)
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
    // ---- per chunk of CH rows: the slab, then the chunk's ticket.  Every thread waits until its slab stores are acknowledged,
    // then ONE atomic per workgroup; the workgroup that draws the last ticket of (shape, chunk) finds all ntile slabs in memory and
    // prepares those rows for the next iteration -- nobody waits for anybody.  The tiles walk the chunks in rotated order, so the
    // chunks of a shape are finished -- and prepared -- by different workgroups side by side (one workgroup summing 25 rows x 32
    // slabs alone: 28 us of a 73 us launch).
    const int nch = (nr + CH - 1) / CH;
    for (int i = 0; i < nch; ++i) {
        const int q = (tile + i) % nch;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int r = grp + NG * u;
            if (r < nr && r / CH == q) {
                float *dst = a.part + (((size_t)b * a.ntile + tile) * a.R + r) * D + 4 * c4;
                st_agent2(dst, acc[u].x, acc[u].y);
                st_agent2(dst + 2, acc[u].z, acc[u].w);
            }
        }
        if (MSR_PROBE == 1) continue;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int *ticket = a.counter + b * NCH_CAP + q;
            const int last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ntile - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next launch)
            s_last = last;
        }
        __syncthreads();
        if (s_last) prep_rows<D>(a, b, q * CH, min(nr, q * CH + CH), nullptr);
    }
}

// grid (ntile, B), 256 threads: dX_j += sum_t sum_r gS^t_rj z^t_r + K^t_rj gO^t_r, + g0_r where key j is kept point r
template <int D, int RMAX>
__global__ __launch_bounds__(256) void ms_rows_apply_kernel(RowsArgs a)
{
    constexpr int ND4 = D / 4;
    constexpr int NG = 256 / ND4;          // key groups (8 at D = 128)
    constexpr int KPG = KT / NG;           // keys per thread
    __shared__ __attribute__((aligned(16))) float s_gs[RMAX][KT], s_k[RMAX][KT];
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nr = live_rows(a, b);
    if (nr == 0) return;
    const int k0 = tile * KT;
    const int c4 = tid % ND4, grp = tid / ND4;
    // (tried: two-wide packed FMAs, v_pk_fma_f32 -- 134 -> 192 us at 25 live rows: the broadcast operands cost more moves than the
    // packing saves)
    float4 acc[KPG];
#pragma unroll
    for (int q = 0; q < KPG; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = a.T - 1; t >= 0; --t) {
        const float *cg = a.all_gs + ((size_t)t * a.B + b) * a.R * a.N, *ck = a.all_k + ((size_t)t * a.B + b) * a.R * a.N;
        __syncthreads();
        for (int e = tid; e < nr * KT; e += 256) {
            const int r = e >> 6, jj = e & 63;
            const bool in = k0 + jj < a.N;
            s_gs[r][jj] = in ? cg[(size_t)r * a.N + k0 + jj] : 0.f;
            s_k[r][jj] = in ? ck[(size_t)r * a.N + k0 + jj] : 0.f;
        }
        __syncthreads();
        const float *zr = a.all_zrow + ((size_t)t * a.B + b) * a.R * D + 4 * c4;
        const float *go = a.all_gO + ((size_t)t * a.B + b) * a.R * D + 4 * c4;
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
    }
    // Z_0 = X.clone() (:60): the kept rows of iterate 0 are rows of X (two slots may name one point: added in slot order)
    for (int r = 0; r < nr; ++r) {
        const int kl = (int)row_id(a, b, r) - k0 - grp * KPG;
        if (kl >= 0 && kl < KPG) {
            const float4 g = *reinterpret_cast<const float4 *>(a.g0 + ((size_t)b * a.R + r) * D + 4 * c4);
#pragma unroll
            for (int q = 0; q < KPG; ++q)
                if (q == kl) { acc[q].x += g.x; acc[q].y += g.y; acc[q].z += g.z; acc[q].w += g.w; }
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

struct RowsLayout {
    size_t rows, grs, g0, part, coef, counter, total;   // offsets in floats (rows / coef: per iteration strides), total
    size_t rows_stride, grs_stride, coef_stride;
};

RowsLayout rows_layout(int B, int N, int D, int R, int T)
{
    const size_t pad = 3;
    RowsLayout l;
    const size_t ntile = (N + KT - 1) / KT, Tn = T > 0 ? T : 1;
    l.rows_stride = (size_t)B * R * D;
    l.grs_stride = ((size_t)B * R + pad) & ~pad;
    l.coef_stride = ((size_t)B * R * N + pad) & ~pad;
    size_t at = 0;
    l.rows = at; at += 2 * Tn * l.rows_stride;           // z rows [T], then gO rows [T]
    l.grs = at; at += Tn * l.grs_stride;
    l.g0 = at; at += l.rows_stride;
    l.part = at; at += (size_t)B * ntile * R * D;
    l.coef = at; at += 2 * Tn * l.coef_stride;           // gS tables [T], then K tables [T]
    l.counter = at; at += ((size_t)B * NCH_CAP + pad) & ~pad;
    l.total = at;
    return l;
}

template <int D, int RMAX>
int rows_bwd(const float *X, const float *bw, int B, int N, int T, const float *const *Zin, const float *const *Zout,
             const float *const *O, const float *const *rsum, const float *const *nrm, const long long *ids,
             const int *nrows, int R, const float *g_rows, float *ws, float *dX, hipStream_t st)
{
    for (int t = 0; t < T; ++t)
        if (!Zin[t] || !Zout[t] || !O[t] || !rsum[t] || !nrm[t]) return PRIFIT_EINVAL;
    const RowsLayout l = rows_layout(B, N, D, R, T);
    const size_t Tn = T > 0 ? T : 1;
    RowsArgs a = {};
    a.X = X; a.bw = bw; a.ids = ids; a.nrows = nrows; a.dX = dX; a.B = B; a.N = N; a.R = R; a.T = T;
    a.ntile = (N + KT - 1) / KT;
    float *zrows = ws + l.rows, *gorows = ws + l.rows + Tn * l.rows_stride, *grs = ws + l.grs;
    float *cgs = ws + l.coef, *ck = ws + l.coef + Tn * l.coef_stride;
    a.g0 = ws + l.g0; a.part = ws + l.part; a.counter = reinterpret_cast<int *>(ws + l.counter);
    a.all_zrow = zrows; a.all_gO = gorows; a.all_gs = cgs; a.all_k = ck;
    auto prepares = [&](int t) {                         // what prep_rows writes next: iteration t, or iterate 0 (t < 0)
        if (t >= 0) {
            a.pZin = Zin[t]; a.pZout = Zout[t]; a.pO = O[t]; a.prsum = rsum[t]; a.pnrm = nrm[t];
            a.p_zrow = zrows + (size_t)t * l.rows_stride; a.p_gO = gorows + (size_t)t * l.rows_stride;
            a.p_grs = grs + (size_t)t * l.grs_stride;
        } else {
            a.pZin = a.pZout = a.pO = a.prsum = a.pnrm = nullptr;
            a.p_zrow = a.p_gO = a.p_grs = nullptr;
        }
    };
    a.g_rows = g_rows;
    prepares(T - 1);
    hipLaunchKernelGGL(ms_rows_first_kernel<D>, dim3(B), dim3(256), 0, st, a);
    for (int t = T - 1; t >= 0; --t) {
        prepares(t - 1);
        hipLaunchKernelGGL((ms_rows_iter_kernel<D, RMAX>), dim3(a.ntile, B), dim3(256), 0, st, a, zrows + (size_t)t * l.rows_stride,
                           gorows + (size_t)t * l.rows_stride, grs + (size_t)t * l.grs_stride, cgs + (size_t)t * l.coef_stride,
                           ck + (size_t)t * l.coef_stride);
    }
    hipLaunchKernelGGL((ms_rows_apply_kernel<D, RMAX>), dim3(a.ntile, B), dim3(256), 0, st, a);
    return prifit_check_launch();
}

}  // namespace

extern "C" {

int prifit_meanshift_rows_supported(int N, int D, int R)
{
    return (N > 0 && R >= 1 && R <= RCAP && (D == 32 || D == 64 || D == 128)) ? 1 : 0;
}

long long prifit_meanshift_rows_bwd_workspace(int B, int N, int D, int R, int T)
{
    if (B <= 0 || T < 0 || !prifit_meanshift_rows_supported(N, D, R)) return 0;
    return (long long)rows_layout(B, N, D, R, T).total;
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
#define ROWS_CASE(DD)                                                                                                  \
    return R <= 32 ? rows_bwd<DD, 32>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, st) \
                   : rows_bwd<DD, 64>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, st)
    switch (D) {
    case 128: ROWS_CASE(128);
    case 64: ROWS_CASE(64);
    default: ROWS_CASE(32);
    }
#undef ROWS_CASE
}

}  // extern "C"
