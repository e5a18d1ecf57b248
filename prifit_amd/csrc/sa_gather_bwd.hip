// First layer of a set-abstraction MLP written by linearity, y1[(g, k)] = (U[idx[g, k]] - Vc[g]) + bias (models/pointnet_util.py
// :243-252 through the first conv; csrc/sa_group.hip MODE 1), backward WITHOUT atomics and without reading y1 (round 5, late):
//     dY = a (y s + t > 0 ? G : 0) + (b y + d)          (bn_relu_bwd_apply_kernel's expression; G = gradient w.r.t. relu(bn(y1)))
//     dU[b, n]  = sum over the in-edges e = (g, k) of point n of dY[e]        dVc[b, g] = -sum_k dY[(g, k)]
// The walk-and-stage kernel of csrc/bn.hip (gather_linear_bwd_bn_kernel) reads G and y1 once (0.6 GB at SA2) but spends most of
// its 0.36 ms per step scanning every index list for the rows of its point range, staging in LDS and flushing with float
// atomics.  Two things make a plain gather possible:
//   * all in-edges of a point carry the SAME row U[n], so y1 = (U[n] - Vc[g]) + bias is re-formed from a [S, C] table that sits in
//     the L2 (the same subtraction the forward did: the same bits) -- the gather reads ONLY the rows of G;
//   * the in-edge lists come from a CSR of the ball-query lists (prifit_list_csr: the counting sort of the DGCNN edge block).
// dU: the lists cut into chunks of 64 entries, one wave each (ball queries pad with their FIRST index, so a few low-index points
// own thousands of list entries: one wave per point took 287 us per launch, all of it in those waves); lane = two channels, 8 rows
// of G in flight, sums in fp64 (the order inside a CSR list differs from run to run; in fp64 that moves the fp32 result only on
// a double-rounding tie); a list that spans chunks is summed from the chunks' partials in chunk order by a second small pass.
// (Measured and dropped: the chunk's entries sorted by (point, entry) before the walk, so that padded runs read contiguous rows --
// 55.5 -> 58.4 us per launch; 4 / 8 / 16 rows in flight: 53.4 / 55.3 / 55.8.)  dVc: one wave per centre over its K contiguous
// rows of G, the U rows from the L2, fp32 sums in slot order.  Each pass reads G once.
#include "common.h"

namespace {

#ifndef GB_UNR_N
#define GB_UNR_N 8
#endif
constexpr int GB_UNR = GB_UNR_N;

struct GatherBwdArgs {
    const float *G, *U, *Vc, *bias, *scale, *shift, *ca, *cb, *cd;
    const int32_t *idx, *offs, *lst, *owner;
    int B, N, S, K, C, nchunk;
    float *dU, *dVc;
    double *part;                      // [B][nchunk][2 (head, tail)][C][2 (sum masked G, sum y1)]
};

__device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }

// grid (nchunk, B): one wave per chunk of 64 list positions.  The entries of a chunk belong to a few consecutive points; the sums
// of a point whose whole list lies inside the chunk go straight to dU, the others (a list that began before the chunk: head
// slot; one that goes on after it: tail slot) to the chunk's partial slots.
__global__ __launch_bounds__(256) void gather_bwd_points_kernel(const GatherBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (chunk >= a.nchunk) return;
    const int C = a.C, c = 2 * lane;
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    const int E = a.S * a.K;
    const int32_t *ob = a.offs + (size_t)b * (a.N + 1);
    const int total = __builtin_amdgcn_readfirstlane(ob[a.N]);           // list entries of this shape
    const int p0 = chunk * 64;
    if (p0 >= total) return;
    const int cnt = min(64, total - p0);
    const float2 sc = ld2(a.scale + cc), sh = ld2(a.shift + cc), ca = ld2(a.ca + cc), cb = ld2(a.cb + cc), cd = ld2(a.cd + cc);
    const float2 bi = a.bias ? ld2(a.bias + cc) : make_float2(0.f, 0.f);
    const int el = lane < cnt ? a.lst[(size_t)b * E + p0 + lane] : 0;
    const int nl = lane < cnt ? a.owner[(size_t)b * E + p0 + lane] : -1;
    const float *Gb = a.G + (size_t)b * E * C + cc, *Vb = a.Vc + (size_t)b * a.S * C + cc, *Ub = a.U + (size_t)b * a.N * C + cc;
    double gx = 0.0, gy = 0.0, yx = 0.0, yy = 0.0;
    int cur = __builtin_amdgcn_readlane(nl, 0);
    float2 u = ld2(Ub + (size_t)cur * C);
    auto flush = [&](int n) {
        const int s0 = ob[n], s1 = ob[n + 1];                            // (wave-uniform loads)
        if (s0 >= p0 && s1 <= p0 + 64) {
            if (cok) {
                const double deg = (double)(s1 - s0);
                float2 o;
                o.x = (float)((double)ca.x * gx + (double)cb.x * yx + (double)cd.x * deg);
                o.y = (float)((double)ca.y * gy + (double)cb.y * yy + (double)cd.y * deg);
                *reinterpret_cast<float2 *>(a.dU + ((size_t)b * a.N + n) * C + c) = o;
            }
        } else if (cok) {
            double *dst = a.part + ((((size_t)b * a.nchunk + chunk) * 2 + (s0 < p0 ? 0 : 1)) * C + c) * 2;
            dst[0] = gx; dst[1] = yx; dst[2] = gy; dst[3] = yy;
        }
    };
    for (int t = 0; t < cnt; t += GB_UNR) {
        float2 gv[GB_UNR], vv[GB_UNR];
        int nn[GB_UNR];
#pragma unroll
        for (int j = 0; j < GB_UNR; ++j) {
            const bool live = t + j < cnt;                     // (wave-uniform)
            // unconditional loads (a dead slot re-reads the batch's first row): a branch around a load makes the compiler wait for
            // it at the join and the rows of a batch would go out one by one
            const int e = __builtin_amdgcn_readlane(el, live ? t + j : t);
            nn[j] = live ? __builtin_amdgcn_readlane(nl, t + j) : -1;
            gv[j] = ld2(Gb + (size_t)e * C);
            vv[j] = ld2(Vb + (size_t)(e / a.K) * C);
        }
#pragma unroll
        for (int j = 0; j < GB_UNR; ++j) {
            if (nn[j] >= 0) {                                  // (wave-uniform)
                if (nn[j] != cur) {
                    flush(cur);
                    cur = nn[j];
                    u = ld2(Ub + (size_t)cur * C);
                    gx = gy = yx = yy = 0.0;
                }
                const float y0 = (u.x - vv[j].x) + bi.x, y1 = (u.y - vv[j].y) + bi.y;      // y1 of the forward, bit for bit
                gx += fmaf(y0, sc.x, sh.x) > 0.f ? (double)gv[j].x : 0.0;
                gy += fmaf(y1, sc.y, sh.y) > 0.f ? (double)gv[j].y : 0.0;
                yx += (double)y0; yy += (double)y1;
            }
        }
    }
    flush(cur);
}

// grid (N / 4, B): one wave per point; those whose list spans chunks are summed from the chunks' slots in chunk order, those with
// no entry at all get their zero row.  (A list inside one chunk was finished by the pass above.)
__global__ __launch_bounds__(256) void gather_bwd_combine_kernel(const GatherBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (n >= a.N) return;
    const int C = a.C, c = 2 * lane;
    if (c >= C) return;
    const int32_t *ob = a.offs + (size_t)b * (a.N + 1);
    const int s0 = ob[n], s1 = ob[n + 1];
    float2 *dst = reinterpret_cast<float2 *>(a.dU + ((size_t)b * a.N + n) * C + c);
    if (s1 == s0) { *dst = make_float2(0.f, 0.f); return; }
    const int c0 = s0 >> 6, c1 = (s1 - 1) >> 6;
    if (c0 == c1) return;
    double gx = 0.0, gy = 0.0, yx = 0.0, yy = 0.0;
    for (int ch = c0; ch <= c1; ++ch) {
        const double *src = a.part + ((((size_t)b * a.nchunk + ch) * 2 + ((s0 < ch * 64) ? 0 : 1)) * C + c) * 2;
        gx += src[0]; yx += src[1]; gy += src[2]; yy += src[3];
    }
    const float2 ca = ld2(a.ca + c), cb = ld2(a.cb + c), cd = ld2(a.cd + c);
    const double deg = (double)(s1 - s0);
    *dst = make_float2((float)((double)ca.x * gx + (double)cb.x * yx + (double)cd.x * deg),
                       (float)((double)ca.y * gy + (double)cb.y * yy + (double)cd.y * deg));
}

__global__ __launch_bounds__(256) void gather_bwd_centres_kernel(const GatherBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)a.B * a.S) return;
    const int b = (int)(gid / a.S);
    const int C = a.C, c = 2 * lane;
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    const float2 sc = ld2(a.scale + cc), sh = ld2(a.shift + cc), ca = ld2(a.ca + cc), cb = ld2(a.cb + cc), cd = ld2(a.cd + cc);
    const float2 bi = a.bias ? ld2(a.bias + cc) : make_float2(0.f, 0.f);
    const float2 vc = ld2(a.Vc + (size_t)gid * C + cc);
    const int32_t *ix = a.idx + (size_t)gid * a.K;
    const float *Gg = a.G + (size_t)gid * a.K * C + cc, *Ub = a.U + (size_t)b * a.N * C + cc;
    float2 acc = make_float2(0.f, 0.f);
    for (int k0 = 0; k0 < a.K; k0 += 64) {
        const int cnt = min(64, a.K - k0);
        const int nl = lane < cnt ? ix[k0 + lane] : -1;
        for (int t = 0; t < cnt; t += GB_UNR) {
            float2 gv[GB_UNR], uv[GB_UNR];
            bool ok[GB_UNR];
#pragma unroll
            for (int j = 0; j < GB_UNR; ++j) {
                const bool live = t + j < cnt;
                const int nn = __builtin_amdgcn_readlane(nl, live ? t + j : t);
                ok[j] = live && nn >= 0 && nn < a.N;           // an index outside [0, N): a constant row, no gradient to anybody
                gv[j] = ld2(Gg + (size_t)(k0 + (live ? t + j : t)) * C);
                uv[j] = ld2(Ub + (size_t)(ok[j] ? nn : 0) * C);
            }
#pragma unroll
            for (int j = 0; j < GB_UNR; ++j) {
                if (ok[j]) {
                    const float y0 = (uv[j].x - vc.x) + bi.x, y1 = (uv[j].y - vc.y) + bi.y;
                    acc.x += fmaf(ca.x, fmaf(y0, sc.x, sh.x) > 0.f ? gv[j].x : 0.f, fmaf(cb.x, y0, cd.x));
                    acc.y += fmaf(ca.y, fmaf(y1, sc.y, sh.y) > 0.f ? gv[j].y : 0.f, fmaf(cb.y, y1, cd.y));
                }
            }
        }
    }
    if (cok) *reinterpret_cast<float2 *>(a.dVc + (size_t)gid * C + c) = make_float2(-acc.x, -acc.y);
}

}  // namespace

extern "C" {

int prifit_gather_linear_bwd_csr_supported(int N, int C) { return (N > 0 && N <= 8192 && C > 0 && C <= 128 && C % 2 == 0) ? 1 : 0; }

long long prifit_gather_linear_bwd_csr_workspace(int B, int S, int K, int C)
{
    if (B <= 0 || S <= 0 || K <= 0 || C <= 0) return 0;
    const long long nchunk = ((long long)S * K + 63) / 64;
    return (long long)B * nchunk * 2 * C * 2;
}

int prifit_gather_linear_bwd_csr(const float *G, const float *U, const float *Vc, const float *bias, const float *scale,
                                 const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                                 const int32_t *idx, const int32_t *offs, const int32_t *lst, const int32_t *owner, int B, int N,
                                 int S, int K, int C, float *dU, float *dVc, double *workspace, void *stream)
{
    if (!G || !U || !Vc || !scale || !shift || !coef_a || !coef_b || !coef_d || !idx || !offs || !lst || !owner || !dU || !dVc ||
        !workspace || B <= 0 || B > 65535 || S <= 0 || K <= 0 || !prifit_gather_linear_bwd_csr_supported(N, C) ||
        (long long)S * K > 0x7fffffffLL || (((uintptr_t)G | (uintptr_t)U | (uintptr_t)Vc | (uintptr_t)dU | (uintptr_t)dVc) & 7) ||
        ((uintptr_t)workspace & 15))
        return PRIFIT_EINVAL;
    GatherBwdArgs a;
    a.G = G; a.U = U; a.Vc = Vc; a.bias = bias; a.scale = scale; a.shift = shift; a.ca = coef_a; a.cb = coef_b; a.cd = coef_d;
    a.idx = idx; a.offs = offs; a.lst = lst; a.owner = owner; a.B = B; a.N = N; a.S = S; a.K = K; a.C = C; a.dU = dU; a.dVc = dVc;
    a.part = workspace; a.nchunk = (int)(((long long)S * K + 63) / 64);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(gather_bwd_points_kernel, dim3((unsigned)((a.nchunk + 3) / 4), (unsigned)B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(gather_bwd_combine_kernel, dim3((unsigned)((N + 3) / 4), (unsigned)B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(gather_bwd_centres_kernel, dim3((unsigned)(((long long)B * S + 3) / 4)), dim3(256), 0, st, a);
    return prifit_check_launch();
}

}  // extern "C"
