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
// Launches (round 6: THREE -- rows_first, ms_rows_queue_kernel = every iteration, rows_apply; mode 0 keeps round 5's T + 2, one
// rows_iter launch per iteration; before: 2 T + 1, with dX read-modified-written by every iteration):
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
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
    int *failed;                                        // queue mode: set when a wait ran out (rows_apply then writes NaN)
};

// Queue mode (ms_rows_queue_kernel): every iteration in ONE launch.
constexpr int TCAP = 16;                                // iterations one launch can hold (the loss path: 10, src/mean_shift.py:57)
struct QueueTabs {                                      // the saved tensors of every iteration (what `prepares(t)` sets per launch)
    const float *Zin[TCAP], *Zout[TCAP], *O[TCAP], *rsum[TCAP], *nrm[TCAP];
};
struct QueueArgs {
    float *zrows, *gorows, *grs, *cgs, *ck;             // iteration t at + t * stride
    size_t rows_stride, grs_stride, coef_stride;
    int *head;                                          // next item of the queue
    int *ready;                                         // [B][NCH_CAP]: levels whose rows are prepared, per (shape, row chunk)
};
constexpr int WAIT_LIMIT = 1 << 20;                     // polls (~1 us each) before a wait gives up: never reached, see the kernel

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

__device__ __forceinline__ void st_agent(float *p, float x)
{
    __hip_atomic_store(reinterpret_cast<int *>(p), __builtin_bit_cast(int, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
template <int D, bool AGENT = false>
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
            // (AGENT: read by other workgroups of the SAME launch, see ms_rows_queue_kernel)
            if (AGENT) {
                st_agent2(a.p_gO + slot * D + c, gx * rinv, gy * rinv);
                st_agent2(a.p_zrow + slot * D + c, zi.x, zi.y);
            } else {
                *reinterpret_cast<float2 *>(a.p_gO + slot * D + c) = make_float2(gx * rinv, gy * rinv);   // d/dO of O / rowsum
                *reinterpret_cast<float2 *>(a.p_zrow + slot * D + c) = zi;
            }
            gr = -(gx * (oo.x * rinv) + gy * (oo.y * rinv));
        }
        gr = wave_sum_f32(gr);
        if (lane == 0) {                                            // d/d(rowsum); d/dZ through "Z + (Mv - Z)" is exactly 0
            if (AGENT) st_agent(a.p_grs + slot, gr * rinv);
            else a.p_grs[slot] = gr * rinv;
        }
    }
}

// grid (B), 256 threads: the caller's gradient through the last iteration (or, with no iteration, straight to g0); the
// shape's ticket starts at zero
template <int D>
__global__ __launch_bounds__(256) void ms_rows_first_kernel(RowsArgs a)
{
    const int b = blockIdx.x;
    if (threadIdx.x < NCH_CAP) a.counter[b * NCH_CAP + threadIdx.x] = 0;
    if (threadIdx.x < NCH_CAP) a.counter[(a.B + b) * NCH_CAP + threadIdx.x] = 0;     // queue mode: `ready`
    if (b == 0 && threadIdx.x < 2) a.counter[2 * a.B * NCH_CAP + threadIdx.x] = 0;   // queue head, `failed`
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


// ---------------------------------------------------------------------------------------------
// Queue mode: the T iteration launches above as ONE launch (T + 2 launches -> 3).
//
// An iteration needs, per shape, the rows that the LAST workgroup of the previous iteration prepared -- a dependency between
// workgroups.  A grid that simply loops over the iterations and waits would need every workgroup resident at once (768 at B =
// 24: exactly what the chip holds -- and two processes sharing a GPU would starve each other for ever).  Instead the work is a
// QUEUE of items (iteration level, shape, key tile), level-major, claimed with one atomic each: an item waits only for items of
// lower index, and every item of lower index has been claimed by a workgroup that is running -- so the lowest unfinished item
// never waits on anything unfinished, whatever the number of resident workgroups (one is enough).  Every wave leaves when the
// queue is empty.  Shapes are independent: while the last workgroup of shape b sums its slabs, the others are already on the
// next shapes' tiles, and nobody meets at a grid-wide barrier.  What a workgroup computes does not depend on who claims what:
// the result is the per-iteration launches', bit for bit.
//
// Rows prepared INSIDE the launch are written and read as agent-scope accesses (they cross the per-XCD L2s), staged through LDS
// (the per-iteration kernel reads them with scalar loads, which a launch boundary makes safe), the next chunk's rows in flight
// under the current chunk's scores.  The key tile stays in LDS when the next item is the same tile one level on.
// ---------------------------------------------------------------------------------------------
template <int D, int RMAX>
__global__ __launch_bounds__(256) void ms_rows_queue_kernel(RowsArgs a, QueueArgs q, QueueTabs tabs)
{
    constexpr int LDX = D + 4;
    constexpr int DQ = D / 4;
    constexpr int ND4 = D / 4;
    constexpr int NG = 256 / ND4;
    constexpr int NPAIR = RC * D / 2;      // float2 per staged row chunk (256 at D = 128: one per thread)
    static_assert(NPAIR <= 256, "one pair of each row table per thread");
    __shared__ __attribute__((aligned(16))) float s_x[KT * LDX];
    __shared__ float s_p[2][4][RC][KT];
    __shared__ __attribute__((aligned(16))) float s_gs[RMAX][KT];
    __shared__ __attribute__((aligned(16))) float s_rows[2][RC][D];
    __shared__ float s_grs[RC];
    __shared__ int s_item, s_last;

    const int tid = threadIdx.x;
    const int per_level = a.ntile * a.B, total = per_level * a.T;
    const float kmin = __expf(-13.0f);
    int have_b = -1, have_tile = -1;

    for (;;) {
        __syncthreads();
        if (tid == 0) s_item = __hip_atomic_fetch_add(q.head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = s_item;
        if (item >= total) break;                                   // (uniform: every wave of every workgroup ends here)
        const int level = item / per_level, rem = item - level * per_level;
        const int b = rem / a.ntile, tile = rem - b * a.ntile, t = a.T - 1 - level;
        const int nr = live_rows(a, b);
        if (nr == 0) continue;
        const int nch = (nr + CH - 1) / CH;
        const int k0 = tile * KT;
        const float *Xb = a.X + (size_t)b * a.N * D;

        // ---- key tile -> LDS (its loads run under the wait below)
        if (b != have_b || tile != have_tile) {
            for (int i = tid; i < KT * ND4; i += 256) {
                const int j = i / ND4, c4 = i - j * ND4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + j < a.N) v = *reinterpret_cast<const float4 *>(Xb + (size_t)(k0 + j) * D + 4 * c4);
                *reinterpret_cast<float4 *>(s_x + j * LDX + 4 * c4) = v;
            }
            have_b = b; have_tile = tile;
        }
        // ---- the rows of this level are prepared by items of the level before (level 0: by rows_first, a launch earlier)
        if (level > 0 && tid < nch) {
            const int *f = q.ready + b * NCH_CAP + tid;
            int n = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < level) {
                if (++n > WAIT_LIMIT) { __hip_atomic_store(a.failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();

        const float bwv = a.bw[b];
        const float rcp_b2 = 1.0f / (bwv * bwv), c_e2 = rcp_b2 * LOG2E;
        const float *zrow = q.zrows + (size_t)t * q.rows_stride + (size_t)b * a.R * D;
        const float *gO = q.gorows + (size_t)t * q.rows_stride + (size_t)b * a.R * D;
        const float *grs = q.grs + (size_t)t * q.grs_stride + (size_t)b * a.R;
        float *coef_gs = q.cgs + (size_t)t * q.coef_stride, *coef_k = q.ck + (size_t)t * q.coef_stride;

        // ---- scores: thread (key j, column quarter dq = wave), the rows' operands broadcast from LDS
        const int j = tid & 63, dq = __builtin_amdgcn_readfirstlane(tid >> 6);
        float xr[DQ];
#pragma unroll
        for (int i = 0; i < DQ; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(s_x + j * LDX + dq * DQ + i);
            xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
        }
        const int prr = tid / (D / 2), pc = 2 * (tid % (D / 2));       // this thread's pair of a staged chunk
        float2 pz = make_float2(0.f, 0.f), pg = pz;
        float pgrs = 0.f;
        auto fetch = [&](int r0) {
            if (tid < NPAIR && r0 + prr < nr) {
                pz = ld_agent2(zrow + (size_t)(r0 + prr) * D + pc);
                pg = ld_agent2(gO + (size_t)(r0 + prr) * D + pc);
            }
            if (tid < RC && r0 + tid < nr) pgrs = ld_agent(grs + r0 + tid);
        };
        fetch(0);
        for (int r0 = 0; r0 < nr; r0 += RC) {
            const int rn = min(RC, nr - r0);
            if (tid < NPAIR) {
                *reinterpret_cast<float2 *>(&s_rows[0][prr][pc]) = pz;
                *reinterpret_cast<float2 *>(&s_rows[1][prr][pc]) = pg;
            }
            if (tid < RC) s_grs[tid] = pgrs;
            if (r0 + RC < nr) fetch(r0 + RC);
            __syncthreads();
            for (int rr = 0; rr < rn; ++rr) {
                const float *zr = &s_rows[0][rr][dq * DQ], *go = &s_rows[1][rr][dq * DQ];
                float s = 0.f, u = 0.f;
#pragma unroll
                for (int i = 0; i < DQ; i += 4) {
                    const float4 zv = *reinterpret_cast<const float4 *>(zr + i), gv = *reinterpret_cast<const float4 *>(go + i);
                    s = fmaf(zv.x, xr[i], s); s = fmaf(zv.y, xr[i + 1], s); s = fmaf(zv.z, xr[i + 2], s); s = fmaf(zv.w, xr[i + 3], s);
                    u = fmaf(gv.x, xr[i], u); u = fmaf(gv.y, xr[i + 1], u); u = fmaf(gv.z, xr[i + 2], u); u = fmaf(gv.w, xr[i + 3], u);
                }
                s_p[0][dq][rr][j] = s;
                s_p[1][dq][rr][j] = u;
            }
            __syncthreads();
            for (int e = tid; e < rn * KT; e += 256) {
                const int rr = e >> 6, jj = e & 63, r = r0 + rr;
                const float s = (s_p[0][0][rr][jj] + s_p[0][1][rr][jj]) + (s_p[0][2][rr][jj] + s_p[0][3][rr][jj]);
                const float u = (s_p[1][0][rr][jj] + s_p[1][1][rr][jj]) + (s_p[1][2][rr][jj] + s_p[1][3][rr][jj]);
                const float w = fminf(fmaxf(fmaf(s, c_e2, -c_e2), -13.0f * LOG2E), 75.0f * LOG2E);
                const float kv = (k0 + jj < a.N) ? __builtin_amdgcn_exp2f(w) : 0.f;
                const float gs = kv > kmin ? (u + s_grs[rr]) * kv * rcp_b2 : 0.f;
                s_gs[r][jj] = gs;
                if (k0 + jj < a.N) {
                    const size_t at = ((size_t)b * a.R + r) * a.N + k0 + jj;
                    coef_gs[at] = gs;
                    coef_k[at] = kv;
                }
            }
            __syncthreads();
        }

        // ---- partial dZ_r = sum_j gS_rj x_j (as ms_rows_iter_kernel)
        const int c4 = tid % ND4, grp = tid / ND4;
        constexpr int RPT = RMAX / NG > 0 ? RMAX / NG : 1;
        float4 acc[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int jj = 0; jj < KT; jj += 4) {
            float4 xv[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) xv[qq] = *reinterpret_cast<const float4 *>(s_x + (jj + qq) * LDX + 4 * c4);
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
        // ---- slabs, tickets; the last workgroup of (shape, chunk) prepares the chunk's rows for the next level and says so
        RowsArgs p = a;
        if (t > 0) {
            p.pZin = tabs.Zin[t - 1]; p.pZout = tabs.Zout[t - 1]; p.pO = tabs.O[t - 1]; p.prsum = tabs.rsum[t - 1]; p.pnrm = tabs.nrm[t - 1];
            p.p_zrow = q.zrows + (size_t)(t - 1) * q.rows_stride; p.p_gO = q.gorows + (size_t)(t - 1) * q.rows_stride;
            p.p_grs = q.grs + (size_t)(t - 1) * q.grs_stride;
        } else {
            p.pZin = p.pZout = p.pO = p.prsum = p.pnrm = nullptr;
            p.p_zrow = p.p_gO = p.p_grs = nullptr;
        }
        for (int i = 0; i < nch; ++i) {
            const int qc = (tile + i) % nch;
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const int r = grp + NG * u;
                if (r < nr && r / CH == qc) {
                    float *dst = a.part + (((size_t)b * a.ntile + tile) * a.R + r) * D + 4 * c4;
                    st_agent2(dst, acc[u].x, acc[u].y);
                    st_agent2(dst + 2, acc[u].z, acc[u].w);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int *ticket = a.counter + b * NCH_CAP + qc;
                const int last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ntile - 1;
                if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next level: nobody
                s_last = last;                                                                         //  draws before `ready` says so)
            }
            __syncthreads();
            if (s_last) {
                prep_rows<D, true>(p, b, qc * CH, min(nr, qc * CH + CH), nullptr);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(q.ready + b * NCH_CAP + qc, level + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Chain mode: one workgroup per (shape, live row) runs ALL T iterations of that row -- TWO launches in all (this + rows_apply).
//
// Row r of an iterate depends on row r of the previous iterate alone -- forward AND backward: dZ_r = sum_j gS_rj x_j needs the
// row's own (gO_r, g_rowsum_r, z_r) and the dictionary, nothing of the other rows.  Splitting an iteration over KEY tiles (above)
// makes every iteration a reduction across workgroups: ~7 dependent memory round trips of ~2 us per iteration (slabs, ticket,
// slab sums, rows, flag), 20-24 us where the arithmetic is ~2.  Splitting over ROWS there is nothing to hand over: a workgroup
// streams the shape's dictionary (N x D, 1 MB at N = 2048: L2 resident, the workgroups of a shape share one XCD) once per
// iteration, 512 threads = 64 key slots x 8 column groups (an "oct" reads one whole 128-byte line per load instruction), the
// row's vectors replicated in registers, sums over D across the oct by DPP, the sum over keys through LDS once per iteration in
// a fixed order.  No atomics, no agent-scope traffic, no tickets; the K and gS tables for rows_apply are written as before.
// Cost: 6 N D flops per row and iteration on ONE CU: 13.7 us per iteration at N = 2048, D = 128 whatever the number of live
// rows up to one per CU (measured: 137 / 140 / 148 us for 1 / 4 / 8 live rows per shape at B = 24, T = 10, against 151 / 166 /
// 198 us for the queue kernel and 10 x 12 / 16 / 22 us for the per-iteration launches); from ~16 live rows per shape (384
// workgroups on 256 CUs) the key-tiled forms, which read the dictionary once for all rows, are as fast (287 vs 280 us) -- the
// caller chooses (prifit_meanshift_rows_bwd `mode`).
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// over the 8 lanes of an oct (lanes 8 k .. 8 k + 7), every lane gets the sum: on the DPP network (`__shfl_xor` is a
// ds_bpermute round trip through the LDS pipe -- six of them in a row per pass of 64 keys)
__device__ __forceinline__ float oct_sum(float v)
{
    v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);    // row_half_mirror: lane i <-> 7 - i of its half row, i.e. the other quad of the oct
    return v;
}

template <int D>
__global__ __launch_bounds__(512) void ms_rows_chain_kernel(RowsArgs a, QueueArgs q, QueueTabs tabs)
{
    constexpr int NV = D / 32;                 // float4 per thread and key row: columns 32 v + 4 o .. + 4 of oct lane o
    constexpr int KPP = 64;                    // key slots (512 threads / 8)
    constexpr int PD = 3;                      // passes in flight ahead of the one being computed (L2 latency ~ 1 us: 8 waves x
                                               // 3 x 4 KB = 96 KB under way per CU; with one pass ahead the loop ran at half the
                                               // CU's fill rate)
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float s_part[NW][D];
    __shared__ __attribute__((aligned(16))) float s_g[D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = tid & 7, ks = tid >> 3;
    // workgroup -> (shape, row): the rows of a shape on ONE XCD (workgroup i runs on XCD i % 8), row-major so that the live rows
    // (r < nrows[b]) are dispatched first
    const int nb8 = (a.B + 7) / 8;
    const int rest = blockIdx.x >> 3;
    const int b = (rest % nb8) * 8 + (blockIdx.x & 7), r = rest / nb8;
    if (b >= a.B || r >= live_rows(a, b)) return;

    const float bwv = a.bw[b];
    const float rcp_b2 = 1.0f / (bwv * bwv), c_e2 = rcp_b2 * LOG2E;
    const float kmin = __expf(-13.0f);
    const size_t slot = (size_t)b * a.R + r;
    const size_t row = (size_t)b * a.N + row_id(a, b, r);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.X + (size_t)b * a.N * D), 0,
                                                                         a.N * D * 4, 0x00020000);
    const int npass = ((a.N + KPP - 1) / KPP + PD) / (PD + 1) * (PD + 1);     // whole rounds of the PD + 1 buffers

    float4 g[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) g[v] = *reinterpret_cast<const float4 *>(a.g_rows + slot * D + 32 * v + 4 * o);

    // the saved rows of an iteration do not depend on the gradient: those of iteration t - 1 are loaded under the keys of t
    float4 zo[NV], oo[NV], z[NV];
    float rs, nm;
    auto load_rows = [&](int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            zo[v] = *reinterpret_cast<const float4 *>(tabs.Zout[t] + row * D + 32 * v + 4 * o);
            oo[v] = *reinterpret_cast<const float4 *>(tabs.O[t] + row * D + 32 * v + 4 * o);
            z[v] = *reinterpret_cast<const float4 *>(tabs.Zin[t] + row * D + 32 * v + 4 * o);
        }
        rs = tabs.rsum[t][row]; nm = tabs.nrm[t][row];
    };
    load_rows(a.T - 1);

    for (int t = a.T - 1; t >= 0; --t) {
        // ---- through the update out = normalize(Z + (O / rowsum - Z)) of iteration t (prep_rows)
        float4 zr[NV], go[NV];
        float grs;
        {
            const float rinv = 1.0f / rs, ninv = 1.0f / nm;
            float dot = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) dot += (g[v].x * zo[v].x + g[v].y * zo[v].y) + (g[v].z * zo[v].z + g[v].w * zo[v].w);
            dot = oct_sum(dot);
            float gr = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float gx = (g[v].x - zo[v].x * dot) * ninv, gy = (g[v].y - zo[v].y * dot) * ninv;
                const float gz = (g[v].z - zo[v].z * dot) * ninv, gw = (g[v].w - zo[v].w * dot) * ninv;
                go[v] = make_float4(gx * rinv, gy * rinv, gz * rinv, gw * rinv);
                gr -= (gx * (oo[v].x * rinv) + gy * (oo[v].y * rinv)) + (gz * (oo[v].z * rinv) + gw * (oo[v].w * rinv));
                zr[v] = z[v];
            }
            grs = oct_sum(gr) * rinv;
        }
        if (ks == 0) {                             // the rows' tables for rows_apply
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                *reinterpret_cast<float4 *>(q.zrows + (size_t)t * q.rows_stride + slot * D + 32 * v + 4 * o) = zr[v];
                *reinterpret_cast<float4 *>(q.gorows + (size_t)t * q.rows_stride + slot * D + 32 * v + 4 * o) = go[v];
            }
        }
        if (t > 0) load_rows(t - 1);
        const __amdgpu_buffer_rsrc_t gs_rs = __builtin_amdgcn_make_buffer_rsrc(q.cgs + (size_t)t * q.coef_stride + slot * a.N, 0, a.N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(q.ck + (size_t)t * q.coef_stride + slot * a.N, 0, a.N * 4, 0x00020000);

        // ---- the keys: s = z . x_j, u = gO . x_j, K, gS; acc += gS x_j
        float4 acc[NV], xb[PD + 1][NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        // (buffer loads: a scalar offset per pass + ONE loop-invariant lane offset, keys beyond N read as zeros without a branch
        // -- a branch around a load makes the compiler wait for the loads before it, and the passes in flight are gone)
        const int off0 = (ks * D + 4 * o) * 4;
        auto load = [&](float4 (&x)[NV], int key) {
            const int soff = (key - ks) * D * 4;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                x[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off0 + 128 * v, soff, 0));
        };
        // (sched_barrier: the loads leave in pass order -- the wait for a pass is "all but the N newest", and a pass whose loads
        // were issued last would make that wait a wait for everything)
#pragma unroll
        for (int i = 0; i < PD; ++i) { load(xb[i], i * KPP + ks); __builtin_amdgcn_sched_barrier(0); }
        for (int p = 0; p < npass; p += PD + 1) {
#pragma unroll
            for (int i = 0; i <= PD; ++i) {
                const int key = (p + i) * KPP + ks;
                load(xb[(i + PD) % (PD + 1)], key + PD * KPP);          // (beyond N: zeros, no access)
                __builtin_amdgcn_sched_barrier(0);
                const float4 (&xc)[NV] = xb[i];
                // (two-wide packed FMAs, v_pk_fma_f32: half the instructions; measured the same 13.7 us per iteration as scalar
                // FMAs, and the same with 1 or 3 passes in flight and with 1 or 8 live rows per shape: what an iteration costs is
                // the serial chain of one workgroup -- 32 passes of load -> 16 dependent FMAs -> DPP sums -> exp -> FMAs on two waves
                // per SIMD -- not bandwidth; profiles/r06_ms_rows.txt)
                f32x2 s2 = {0.f, 0.f}, u2 = {0.f, 0.f};
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const f32x2 xl = {xc[v].x, xc[v].y}, xh = {xc[v].z, xc[v].w};
                    s2 = __builtin_elementwise_fma(f32x2{zr[v].x, zr[v].y}, xl, s2);
                    s2 = __builtin_elementwise_fma(f32x2{zr[v].z, zr[v].w}, xh, s2);
                    u2 = __builtin_elementwise_fma(f32x2{go[v].x, go[v].y}, xl, u2);
                    u2 = __builtin_elementwise_fma(f32x2{go[v].z, go[v].w}, xh, u2);
                }
                float s = s2.x + s2.y, u = u2.x + u2.y;
                s = oct_sum(s);
                u = oct_sum(u);
                const float w = fminf(fmaxf(fmaf(s, c_e2, -c_e2), -13.0f * LOG2E), 75.0f * LOG2E);
                const float kv = key < a.N ? __builtin_amdgcn_exp2f(w) : 0.f;
                const float gs = kv > kmin ? (u + grs) * kv * rcp_b2 : 0.f;
                // (buffer stores, the lanes that do not write aimed past the end: a branch here would make every wait on the
                // key loads conservative -- the compiler cannot count the stores of a skipped block)
                const int st_off = (o == 0 && key < a.N) ? key * 4 : 0x7fffffff;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gs), gs_rs, st_off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, kv), k_rs, st_off, 0, 0);
                const f32x2 gs2 = {gs, gs};
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const f32x2 lo = __builtin_elementwise_fma(gs2, f32x2{xc[v].x, xc[v].y}, f32x2{acc[v].x, acc[v].y});
                    const f32x2 hi = __builtin_elementwise_fma(gs2, f32x2{xc[v].z, xc[v].w}, f32x2{acc[v].z, acc[v].w});
                    acc[v] = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
        }
        // ---- dZ_r = the sum over the key slots: the 8 slots of a wave by lane exchange, the waves through LDS in wave order
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            acc[v].x += dpp_f32<0x128>(acc[v].x); acc[v].y += dpp_f32<0x128>(acc[v].y);      // row_ror:8 = lane ^ 8
            acc[v].z += dpp_f32<0x128>(acc[v].z); acc[v].w += dpp_f32<0x128>(acc[v].w);
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) {
                acc[v].x += __shfl_xor(acc[v].x, m); acc[v].y += __shfl_xor(acc[v].y, m);
                acc[v].z += __shfl_xor(acc[v].z, m); acc[v].w += __shfl_xor(acc[v].w, m);
            }
            if (lane < 8) *reinterpret_cast<float4 *>(&s_part[wave][32 * v + 4 * o]) = acc[v];
        }
        __syncthreads();
        if (tid < D) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
            s_g[tid] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) g[v] = *reinterpret_cast<const float4 *>(&s_g[32 * v + 4 * o]);
    }
    // Z_0 = X.clone(): dL/d(row of iterate 0), added to dX at the kept point by rows_apply
    if (ks == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<float4 *>(a.g0 + slot * D + 32 * v + 4 * o) = g[v];
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
    // (queue mode: a wait that ran out -- a GPU so oversubscribed that a claimed item did not finish in ~1 s -- must not pass as
    // a gradient)
    const float poison = (a.failed && *a.failed) ? __builtin_nanf("") : 0.f;
#pragma unroll
    for (int q = 0; q < KPG; ++q) {
        const int key = k0 + grp * KPG + q;
        if (key < a.N) {
            float4 *p = reinterpret_cast<float4 *>(a.dX + ((size_t)b * a.N + key) * D + 4 * c4);
            float4 v = *p;
            v.x += acc[q].x + poison; v.y += acc[q].y + poison; v.z += acc[q].z + poison; v.w += acc[q].w + poison;
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
    l.counter = at; at += ((size_t)2 * B * NCH_CAP + 2 + pad) & ~pad;   // tickets, `ready`, queue head, `failed`
    l.total = at;
    return l;
}

// workgroups of the queue kernel the chip holds at once (more would only queue behind them)
template <int D, int RMAX>
int queue_grid_cap()
{
    static int cap = 0;
    if (cap == 0) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ms_rows_queue_kernel<D, RMAX>, 256, 0) != hipSuccess || per_cu < 1)
            return 256;
        cap = per_cu * p.multiProcessorCount;
    }
    return cap;
}

template <int D, int RMAX>
int rows_bwd(const float *X, const float *bw, int B, int N, int T, const float *const *Zin, const float *const *Zout,
             const float *const *O, const float *const *rsum, const float *const *nrm, const long long *ids,
             const int *nrows, int R, const float *g_rows, float *ws, float *dX, int mode, hipStream_t st)
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
    a.failed = a.counter + 2 * B * NCH_CAP + 1;
    if (mode == 2 && T >= 1 && T <= TCAP) {
        QueueArgs q = {};
        q.zrows = zrows; q.gorows = gorows; q.grs = grs; q.cgs = cgs; q.ck = ck;
        q.rows_stride = l.rows_stride; q.grs_stride = l.grs_stride; q.coef_stride = l.coef_stride;
        QueueTabs tabs = {};
        for (int t = 0; t < T; ++t) { tabs.Zin[t] = Zin[t]; tabs.Zout[t] = Zout[t]; tabs.O[t] = O[t]; tabs.rsum[t] = rsum[t]; tabs.nrm[t] = nrm[t]; }
        a.failed = nullptr;
        hipLaunchKernelGGL(ms_rows_chain_kernel<D>, dim3(8 * ((B + 7) / 8) * R), dim3(512), 0, st, a, q, tabs);
        hipLaunchKernelGGL((ms_rows_apply_kernel<D, RMAX>), dim3(a.ntile, B), dim3(256), 0, st, a);
        return prifit_check_launch();
    }
    prepares(T - 1);
    hipLaunchKernelGGL(ms_rows_first_kernel<D>, dim3(B), dim3(256), 0, st, a);
    if (mode == 1 && T >= 1 && T <= TCAP) {
        QueueArgs q = {};
        q.zrows = zrows; q.gorows = gorows; q.grs = grs; q.cgs = cgs; q.ck = ck;
        q.rows_stride = l.rows_stride; q.grs_stride = l.grs_stride; q.coef_stride = l.coef_stride;
        q.ready = a.counter + B * NCH_CAP; q.head = a.counter + 2 * B * NCH_CAP;
        QueueTabs tabs = {};
        for (int t = 0; t < T; ++t) { tabs.Zin[t] = Zin[t]; tabs.Zout[t] = Zout[t]; tabs.O[t] = O[t]; tabs.rsum[t] = rsum[t]; tabs.nrm[t] = nrm[t]; }
        const long long items = (long long)a.ntile * B;
        const int grid = (int)std::min<long long>(items, queue_grid_cap<D, RMAX>());
        hipLaunchKernelGGL((ms_rows_queue_kernel<D, RMAX>), dim3(grid), dim3(256), 0, st, a, q, tabs);
    } else for (int t = T - 1; t >= 0; --t) {
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
                              const float *g_rows, float *workspace, float *dX, int mode, void *stream)
{
    if (mode < 0 || mode > 2) return PRIFIT_EINVAL;
    if (!X || !bw || !ids || !g_rows || !workspace || !dX || B <= 0 || B > 65535 || T < 0 || (T > 0 && (!Zin || !Zout || !O || !rowsum || !nrm)) ||
        !prifit_meanshift_rows_supported(N, D, R) || (((uintptr_t)X | (uintptr_t)dX | (uintptr_t)workspace) & 15))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
#define ROWS_CASE(DD)                                                                                                  \
    return R <= 32 ? rows_bwd<DD, 32>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, mode, st) \
                   : rows_bwd<DD, 64>(X, bw, B, N, T, Zin, Zout, O, rowsum, nrm, ids, nrows, R, g_rows, workspace, dX, mode, st)
    switch (D) {
    case 128: ROWS_CASE(128);
    case 64: ROWS_CASE(64);
    default: ROWS_CASE(32);
    }
#undef ROWS_CASE
}

}  // extern "C"
