// The DGCNN edge convolution block (src/dgcnn.py:98-107 get_graph_feature + :157-171 conv -> GroupNorm -> LeakyReLU -> max
// over the k neighbours) WITHOUT any per-edge tensor.  By linearity the edge pre-activation is y[(i,j)] = U[idx[i,j]] - Vc[i]
// (U, Vc per POINT, see src/dgcnn.py of the build); everything the block and its autograd need from the B N k rows are
// per-point tables:
//   forward  : per (point, channel) max_j y, min_j y with their first positions, sum_j y, and the GroupNorm column sums --
//              one pass over the neighbour lists (rows of U come from L2: a shape's U is N C floats).  After the statistics
//              are final the pooled activation is act(s y* + t) with y* the maximum (s > 0) or the minimum (s < 0): the
//              activation is monotone in y, so the winner is known without a second pass over the edges.
//   backward : dy[(i,j)] = a T[i] [j == winner] + b y[(i,j)] + d  (the pooled GroupNorm backward, bn.hip), hence
//                dVc[i]  = -(a T[i] + b sum_j y[(i,j)] + k d)             (k = the number of valid neighbours)
//                dU[n]   = a sum_{winners pointing at n} T + b (deg(n) U[n] - sum_{edges (i,j) -> n} Vc[i]) + d deg(n)
//              both sums are GATHERS over the in-edges of n (a CSR of the neighbour lists, built once per graph; which
//              channels an edge won is a 64-bit mask per edge) in place of B N k C float atomics: no atomics at all.
//              (Scattering only the winners with atomics -- one per point and channel -- was 2-4 x slower than the gather:
//              an atomic instruction costs per cache line it touches, and the winners of a point go to ~15 different rows.)
// The [B N k, C] pre-activations (0.25 - 0.5 GB per layer at B = 24, N = 2048, k = 20) are neither written nor read.
// An index outside [0, N) contributes a zero row (as prifit_gather_linear_fwd does) and receives no gradient.
#include "common.h"

namespace {

constexpr int EC_PTS = 16;       // points per workgroup of the statistics pass = one GroupNorm slab (4 per wave: the pass is
                                 // latency-bound on the row gathers, so many short waves)
constexpr int EC_MAXN = 8192;    // CSR build: one workgroup per shape, histogram in LDS

__device__ __forceinline__ float act_grad(float z, float g, float slope) { return z > 0.f ? g : g * slope; }

// ---- CSR of the neighbour lists: offs[b][n] .. offs[b][n+1] = positions in lst[b] of the edges (i k + j) with idx[b,i,j] = n;
// pos[b][e] = the position of edge e in lst[b] ----
// SPLIT_FILL: lst[pos[e]] = e is 40 960 scattered 4-byte stores per shape -- 64 cache lines per wave instruction, from the ONE
// CU this workgroup runs on, most of the kernel's 32 us; edge_csr_fill_kernel does them from the whole chip.
template <bool SPLIT_FILL>
__global__ __launch_bounds__(1024) void edge_csr_kernel(const int32_t *__restrict__ idx, int N, int E,
                                                        int32_t *__restrict__ offs, int32_t *__restrict__ lst,
                                                        int32_t *__restrict__ pos)
{
    __shared__ int s_bin[EC_MAXN];
    __shared__ int s_wave[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t *ix = idx + (size_t)b * E;
    for (int i = tid; i < N; i += 1024) s_bin[i] = 0;
    __syncthreads();
#ifndef EC_CSR_CU
#define EC_CSR_CU 8
#endif
    constexpr int CU = EC_CSR_CU;               // independent index loads in flight per thread
    for (int e0 = tid; e0 < E; e0 += 1024 * CU) {
        int n[CU];
#pragma unroll
        for (int j = 0; j < CU; ++j) n[j] = e0 + j * 1024 < E ? ix[e0 + j * 1024] : -1;
#pragma unroll
        for (int j = 0; j < CU; ++j)
            if (n[j] >= 0 && n[j] < N) atomicAdd(&s_bin[n[j]], 1);
    }
    __syncthreads();
    // exclusive scan: each thread owns `per` consecutive bins
    const int per = (N + 1023) / 1024;
    const int lo = min(N, tid * per), hi = min(N, lo + per);
    int local = 0;
    for (int i = lo; i < hi; ++i) local += s_bin[i];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
    int run = base + incl - local;
    int32_t *ob = offs + (size_t)b * (N + 1);
    for (int i = lo; i < hi; ++i) {
        const int c = s_bin[i];
        ob[i] = run;
        s_bin[i] = run;       // becomes the fill cursor
        run += c;
    }
    if (tid == 1023) ob[N] = run;
    __syncthreads();
    int32_t *lb = lst + (size_t)b * E, *pb = pos + (size_t)b * E;
    for (int e0 = tid; e0 < E; e0 += 1024 * CU) {
        int n[CU];
#pragma unroll
        for (int j = 0; j < CU; ++j) n[j] = e0 + j * 1024 < E ? ix[e0 + j * 1024] : -1;
#pragma unroll
        for (int j = 0; j < CU; ++j) {
            const int e = e0 + j * 1024;
            if (e < E) {
                const bool ok = n[j] >= 0 && n[j] < N;
                const int p = ok ? atomicAdd(&s_bin[n[j]], 1) : -1;
                if (ok && !SPLIT_FILL) lb[p] = e;
                pb[e] = p;                                  // where edge e sits in the lists (-1: not in any)
            }
        }
    }
}

// owner (may be NULL) [B, E]: the bin of every list position (the point an in-edge list entry belongs to)
__global__ __launch_bounds__(256) void edge_csr_fill_kernel(const int32_t *__restrict__ pos, const int32_t *__restrict__ idx, int E,
                                                            long long total, int32_t *__restrict__ lst, int32_t *__restrict__ owner)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const int p = pos[id];
        if (p >= 0) {
            lst[(id / E) * E + p] = (int)(id % E);
            if (owner) owner[(id / E) * E + p] = idx[id];
        }
    }
}

// ---- forward pass over the neighbour lists: one wave per point, lane = channels lane, lane + 64, ... ----
template <int V, int UNR>
__global__ __launch_bounds__(256) void edge_stats_kernel(const float *__restrict__ U, long long ldu,
                                                         const float *__restrict__ Vc, long long ldv, int selfterm,
                                                         const int32_t *__restrict__ idx, int B, int N, int k,
                                                         float *__restrict__ ymax, float *__restrict__ ymin,
                                                         int32_t *__restrict__ karg, float *__restrict__ ysum,
                                                         float *__restrict__ vct, float *__restrict__ slab)
{
    constexpr int C = 64 * V;
    __shared__ float s_red[4][2][C];
    int b, blk;
    xcd_shape_block(blockIdx.x, N / EC_PTS, B, b, blk);
    const int p0 = blk * EC_PTS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *Ub = U + (size_t)b * N * ldu + lane;
    float ws[V], wq[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { ws[v] = 0.f; wq[v] = 0.f; }
    const int p_end = min(N, p0 + EC_PTS);
    for (int p = p0 + wave; p < p_end; p += 4) {
        const size_t g = (size_t)b * N + p;
        const int32_t *ix = idx + g * k;
        float vc[V], s[V], mx[V], mn[V];
        int kx[V], kn[V], nv = 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            vc[v] = Vc[g * ldv + lane + 64 * v];
            if (selfterm) vc[v] = U[g * ldu + lane + 64 * v] - vc[v];     // centre term U_i - Vb_i (see prifit_edge_stats)
            vct[g * C + lane + 64 * v] = vc[v];
            s[v] = 0.f; mx[v] = -INFINITY; mn[v] = INFINITY; kx[v] = 0; kn[v] = 0;
        }
        for (int kc = 0; kc < k; kc += 64) {                  // the index list sits in the lanes: no load between two row gathers
            const int kn_here = min(64, k - kc);
            const int nl = lane < kn_here ? ix[kc + lane] : -1;
            for (int kk = 0; kk < kn_here; kk += UNR) {
                float u[UNR][V];
                bool ok[UNR];
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    const int n = kk + j < kn_here ? __builtin_amdgcn_readlane(nl, kk + j) : -1;
                    ok[j] = n >= 0 && n < N;
                    const int nn = ok[j] ? n : 0;            // unconditional loads (see edge_bwd_gather_kernel)
#pragma unroll
                    for (int v = 0; v < V; ++v) u[j][v] = Ub[(size_t)nn * ldu + 64 * v];
                }
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    if (kk + j < kn_here) {
                        nv += ok[j] ? 1 : 0;
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const float y = ok[j] ? u[j][v] - vc[v] : 0.f;
                            s[v] += y;
                            wq[v] = fmaf(y, y, wq[v]);
                            if (y > mx[v]) { mx[v] = y; kx[v] = kc + kk + j; }    // strict: the first maximum / minimum
                            if (y < mn[v]) { mn[v] = y; kn[v] = kc + kk + j; }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const size_t o = g * C + lane + 64 * v;
            ymax[o] = mx[v]; ymin[o] = mn[v]; karg[o] = kx[v] | (kn[v] << 10) | (nv << 20); ysum[o] = s[v];
            ws[v] += s[v];
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) { s_red[wave][0][lane + 64 * v] = ws[v]; s_red[wave][1][lane + 64 * v] = wq[v]; }
    __syncthreads();
    float *sl = slab + ((size_t)b * (N / EC_PTS) + blk) * 2 * C;
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        const int which = t / C, c = t - which * C;
        sl[t] = (s_red[0][which][c] + s_red[1][which][c]) + (s_red[2][which][c] + s_red[3][which][c]);
    }
}

// ---- pooled activation from the tables: out = act(s y* + t), y* kept for the backward ----
__global__ __launch_bounds__(256) void edge_pool_kernel(const float *__restrict__ ymax, const float *__restrict__ ymin,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        int N, int C4, long long total, float slope, float *__restrict__ out,
                                                        long long ldo, float *__restrict__ ystar)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long g = id / C4;
        const int c = (int)(id - g * C4) * 4;
        const long long to = (g / N) * (C4 * 4) + c;
        const float4 s = *reinterpret_cast<const float4 *>(scale + to), t = *reinterpret_cast<const float4 *>(shift + to);
        const float4 hi = *reinterpret_cast<const float4 *>(ymax + g * (C4 * 4) + c);
        const float4 lo = *reinterpret_cast<const float4 *>(ymin + g * (C4 * 4) + c);
        float4 y, o;
        y.x = s.x < 0.f ? lo.x : hi.x; y.y = s.y < 0.f ? lo.y : hi.y; y.z = s.z < 0.f ? lo.z : hi.z; y.w = s.w < 0.f ? lo.w : hi.w;
        const float zx = fmaf(y.x, s.x, t.x), zy = fmaf(y.y, s.y, t.y), zz = fmaf(y.z, s.z, t.z), zw = fmaf(y.w, s.w, t.w);
        o.x = zx > 0.f ? zx : zx * slope; o.y = zy > 0.f ? zy : zy * slope; o.z = zz > 0.f ? zz : zz * slope;
        o.w = zw > 0.f ? zw : zw * slope;
        *reinterpret_cast<float4 *>(out + g * ldo + c) = o;
        *reinterpret_cast<float4 *>(ystar + g * (C4 * 4) + c) = y;
    }
}

// ---- backward, per point (first): aT = a act'(s y* + t) gp, dVc, and for every edge (i, j) the set of channels whose
// winner it is (one 64-bit mask per edge and 64 channels, in the order of the in-edge lists).  One wave per point, lane =
// channels. ----
template <int V>
__global__ __launch_bounds__(256) void edge_bwd_point_kernel(const float *__restrict__ gp, long long ldgp,
                                                             const float *__restrict__ ystar, const float *__restrict__ ysum,
                                                             const int32_t *__restrict__ karg, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, const float *__restrict__ ca,
                                                             const float *__restrict__ cb, const float *__restrict__ cd,
                                                             const int32_t *__restrict__ idx, int N, int k, long long points,
                                                             const int32_t *__restrict__ pos, float slope,
                                                             float *__restrict__ aT, unsigned long long *__restrict__ masks,
                                                             float *__restrict__ dVc, long long ldd, int selfterm)
{
    constexpr int C = 64 * V;
    const int lane = threadIdx.x & 63;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < points; g += (long long)gridDim.x * 4) {
        const int b = (int)(g / N);
        const int32_t *ix = idx + g * k;
        const int nl = lane < k ? ix[lane] : -1;             // the first 64 entries of the list sit in the lanes
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = lane + 64 * v;
            const long long to = (long long)b * C + c;
            const size_t id = (size_t)g * C + c;
            const float s = scale[to], t = shift[to];
            const float T = ca[to] * act_grad(fmaf(ystar[id], s, t), gp[g * ldgp + c], slope);
            const int pk = karg[id];
            const int w = s < 0.f ? ((pk >> 10) & 1023) : (pk & 1023);
            const int n = w < 64 ? __shfl(nl, w, 64) : ix[w];
            // a zero row (index out of range) is a constant: it takes part in the statistics, not in the gradients
            const bool ok = n >= 0 && n < N;
            aT[id] = T;
            const float dvc = -((ok ? T : 0.f) + fmaf(cb[to], ysum[id], (float)(pk >> 20) * cd[to]));
            dVc[(size_t)g * ldd + c] = selfterm ? -dvc : dvc;     // selfterm: the gradient of Vb = U_i - (centre term)
            for (int jc = 0; jc < k; jc += 64) {
                const int cnt = min(64, k - jc);
                unsigned long long mine = 0ull;
                for (int j = 0; j < cnt; ++j) {
                    const unsigned long long m = __ballot(w == jc + j);
                    if (lane == j) mine = m;
                }
                // stored where the edge sits in the in-edge lists: the gather pass reads masks and lists side by side
                const int p = lane < cnt ? pos[(size_t)g * k + jc + lane] : -1;
                if (p >= 0) masks[((size_t)b * N * k + p) * V + v] = mine;
            }
        }
    }
}

// ---- backward, per point n (second): its in-edges from the CSR.  dU[n] = b (deg U[n] - sum Vc[i]) + d deg + sum over the
// in-edges that are winners of aT[i] -- no atomics; the sums in fp64 (the order in which the CSR build filled a list
// differs from run to run; in fp64 that moves the fp32 result only on a double-rounding tie) ----
__global__ __launch_bounds__(256) void edge_bwd_gather_kernel(const float *__restrict__ U, long long ldu,
                                                              const float *__restrict__ Vc, const float *__restrict__ aT,
                                                              const unsigned long long *__restrict__ masks,
                                                              const int32_t *__restrict__ offs,
                                                              const int32_t *__restrict__ lst, const float *__restrict__ cb,
                                                              const float *__restrict__ cd, int B, int N, int k, int C,
                                                              float *__restrict__ dU, long long ldd,
                                                              const float *__restrict__ dVb)
{
    constexpr int UNR = 12;       // in-degrees of a k = 20 graph: median 21, p90 29 -> two or three batches
    const int lane = threadIdx.x & 63;
    const int V = C >> 6, vy = blockIdx.y, c = lane + 64 * vy;       // this wave's 64 channels
    int b, blk;
    xcd_shape_block(blockIdx.x, (N + 3) / 4, B, b, blk);
    const int n = blk * 4 + (threadIdx.x >> 6);
    if (n < N) {
        const size_t g = (size_t)b * N + n;
        const int32_t *ob = offs + (size_t)b * (N + 1);
        const int e0 = __builtin_amdgcn_readfirstlane(ob[n]), e1 = __builtin_amdgcn_readfirstlane(ob[n + 1]);
        const int32_t *lb = lst + (size_t)b * N * k;
        const float *Vb = Vc + (size_t)b * N * C + c, *Ab = aT + (size_t)b * N * C + c;
        const unsigned long long *mb = masks + (size_t)b * N * k * V + vy;
        double acc = 0.0, accw = 0.0;
        for (int ec = e0; ec < e1; ec += 64) {                 // 64 list entries (edge numbers i k + j) and their masks in the lanes
            const int cnt = min(64, e1 - ec);
            const int el = lane < cnt ? lb[ec + lane] : 0;
            const unsigned long long m = lane < cnt ? mb[(size_t)(ec + lane) * V] : 0ull;
            const int il = el / k;
            const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
            for (int t = 0; t < cnt; t += UNR) {
                float x[UNR], z[UNR];
                bool hit[UNR];
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    const bool live = t + j < cnt;          // wave-uniform
                    const int i = live ? __builtin_amdgcn_readlane(il, t + j) : 0;
                    const unsigned lo = live ? (unsigned)__builtin_amdgcn_readlane((int)mlo, t + j) : 0u;
                    const unsigned hi = live ? (unsigned)__builtin_amdgcn_readlane((int)mhi, t + j) : 0u;
                    hit[j] = (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) != 0u;
                    // unconditional loads (row 0 for a dead slot): a branch around a load, even a wave-uniform one, makes the
                    // compiler wait for it at the join and the 2 UNR row gathers of a batch would go out one by one
                    const float xv = Vb[(size_t)i * C], zv = Ab[(size_t)i * C];
                    x[j] = live ? xv : 0.f;
                    z[j] = zv;
                }
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    acc += (double)x[j];
                    accw += hit[j] ? (double)z[j] : 0.0;
                }
            }
        }
        const double deg = (double)(e1 - e0);
        const double u = (double)U[g * ldu + c];
        const double bb = (double)cb[(size_t)b * C + c], d = (double)cd[(size_t)b * C + c];
        // dVb (selfterm): the centre term is U_i - Vb_i, so the point's own row of U also receives dVc_i = -dVb_i
        const double self = dVb ? -(double)dVb[g * ldd + c] : 0.0;
        dU[g * ldd + c] = (float)(bb * (deg * u - acc) + d * deg + accw + self);
    }
}

inline int ew_grid(long long total)
{
    long long g = (total + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > 256 * 32 ? 256 * 32 : g);
}

}  // namespace

extern "C" {

int prifit_edge_points_per_slab(void) { return EC_PTS; }

int prifit_edge_tables_supported(int N, int k, int C)
{
    return (N > 0 && N <= EC_MAXN && N % EC_PTS == 0 && k > 0 && k < 1024 && (C == 64 || C == 128 || C == 256)) ? 1 : 0;
}

int prifit_list_csr(const int32_t *idx, int B, int nbins, int E, int32_t *offs, int32_t *lst, int32_t *pos, int32_t *owner,
                    void *stream);

int prifit_edge_csr(const int32_t *idx, int B, int N, int k, int32_t *offs, int32_t *lst, int32_t *pos, void *stream)
{
    if (!idx || !offs || !lst || !pos || B <= 0 || N <= 0 || N > EC_MAXN || k <= 0 || (long long)N * k > 0x7fffffffLL) return PRIFIT_EINVAL;
    return prifit_list_csr(idx, B, N, N * k, offs, lst, pos, nullptr, stream);
}

int prifit_list_csr(const int32_t *idx, int B, int nbins, int E, int32_t *offs, int32_t *lst, int32_t *pos, int32_t *owner,
                    void *stream)
{
    if (!idx || !offs || !lst || !pos || B <= 0 || nbins <= 0 || nbins > EC_MAXN || E <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(edge_csr_kernel<true>, dim3(B), dim3(1024), 0, as_stream(stream), idx, nbins, E, offs, lst, pos);
    const long long total = (long long)B * E;
    hipLaunchKernelGGL(edge_csr_fill_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), pos, idx, E, total, lst, owner);
    return prifit_check_launch();
}

int prifit_edge_stats(const float *U, long long ldu, const float *Vc, long long ldv, int selfterm, const int32_t *idx, int B, int N,
                      int k, int C, float *ymax, float *ymin, int32_t *karg, float *ysum, float *vct, float *slab, void *stream)
{
    if (!U || !Vc || !idx || !ymax || !ymin || !karg || !ysum || !vct || !slab || B <= 0 || ldu < C || ldv < C ||
        !prifit_edge_tables_supported(N, k, C))
        return PRIFIT_EINVAL;
    const dim3 grid((unsigned)(N / EC_PTS) * B);
    hipStream_t st = as_stream(stream);
#define EDGE_STATS(V, R)                                                                                                      \
    hipLaunchKernelGGL((edge_stats_kernel<V, R>), grid, dim3(256), 0, st, U, ldu, Vc, ldv, selfterm, idx, B, N, k, ymax, ymin, karg, \
                       ysum, vct, slab)
    // row gathers in flight per wave: k = 20 (the reference's default) goes out as two batches of 10
    if (k % 10 == 0) { if (C == 64) EDGE_STATS(1, 10); else if (C == 128) EDGE_STATS(2, 10); else EDGE_STATS(4, 10); }
    else { if (C == 64) EDGE_STATS(1, 8); else if (C == 128) EDGE_STATS(2, 8); else EDGE_STATS(4, 8); }
#undef EDGE_STATS
    return prifit_check_launch();
}

int prifit_edge_pool(const float *ymax, const float *ymin, const float *scale, const float *shift, int B, int N, int C,
                     float slope, float *out, long long ldo, float *ystar, void *stream)
{
    if (!ymax || !ymin || !scale || !shift || !out || !ystar || B <= 0 || N <= 0 || C <= 0 || (C & 3) || ldo < C || (ldo & 3) ||
        ((uintptr_t)out & 15))
        return PRIFIT_EINVAL;
    const long long total = (long long)B * N * (C / 4);
    hipLaunchKernelGGL(edge_pool_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), ymax, ymin, scale, shift, N,
                       C / 4, total, slope, out, ldo, ystar);
    return prifit_check_launch();
}

long long prifit_edge_bwd_workspace(int B, int N, int k, int C)
{
    // aT [B N C] floats, then one 64-bit winner mask per edge and 64 channels
    return (long long)B * N * C * 4 + (long long)B * N * k * (C / 64) * 8;
}

int prifit_edge_bwd(const float *gp, long long ldgp, const float *ystar, const float *ysum, const int32_t *karg,
                    const float *scale, const float *shift, const float *ca, const float *cb, const float *cd, const float *U,
                    long long ldu, const float *vct, int selfterm, const int32_t *idx, const int32_t *offs, const int32_t *lst,
                    const int32_t *pos, int B, int N, int k, int C, float slope, float *dU, float *dVc, long long ldd,
                    void *workspace, void *stream)
{
    if (!gp || !ystar || !ysum || !karg || !scale || !shift || !ca || !cb || !cd || !U || !vct || !idx || !offs || !lst || !pos ||
        !dU || !dVc || !workspace || ((uintptr_t)workspace & 7) || B <= 0 || ldgp < C || ldu < C || ldd < C ||
        !prifit_edge_tables_supported(N, k, C))
        return PRIFIT_EINVAL;
    const long long points = (long long)B * N;
    hipStream_t st = as_stream(stream);
    const int grid = (int)((points + 3) / 4 > 256 * 64 ? 256 * 64 : (points + 3) / 4);
    float *aT = static_cast<float *>(workspace);
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(aT + points * C);
#define EDGE_POINT(V)                                                                                                        \
    hipLaunchKernelGGL(edge_bwd_point_kernel<V>, dim3(grid), dim3(256), 0, st, gp, ldgp, ystar, ysum, karg, scale, shift, ca, cb, \
                       cd, idx, N, k, points, pos, slope, aT, masks, dVc, ldd, selfterm)
    if (C == 64) EDGE_POINT(1);
    else if (C == 128) EDGE_POINT(2);
    else EDGE_POINT(4);
#undef EDGE_POINT
    hipLaunchKernelGGL(edge_bwd_gather_kernel, dim3((unsigned)((N + 3) / 4) * B, C / 64), dim3(256), 0, st, U, ldu, vct, aT, masks,
                       offs, lst, cb, cd, B, N, k, C, dU, ldd, selfterm ? (const float *)dVc : (const float *)nullptr);
    return prifit_check_launch();
}

}  // extern "C"
