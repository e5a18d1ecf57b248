// Max-pooled last layer of a shared MLP, backward in the ALGEBRAIC form (models/pointnet_util.py:252-256 through autograd;
// round 5): the rows that are NOT row-dense.
//
// With dY = T [row == winner] + b Y + d and Y = A W^T + bias (A = relu(bn(Yp)) the layer's input, T [G, Cout] the pooled
// gradient through BatchNorm + ReLU at each group's winning row):
//     dA = A M + 1 v^T  +  S W                     S [P, Cout]: T at (winner row, channel), zero elsewhere
//     dW = diag(b) W (A^T A) + (d + b * bias) (1^T A)  +  S^T A
// The first terms are prifit_pool_alg_dense_f32 (csrc/gemm_stream_bwd.hip).  This file adds the S terms -- Cout entries per
// group of K rows -- as index work on the vector ALU:
//   rows:     for every row of a group that won channels {c}:  Gp[row] += sum_c T_c W[c, :]  (and its share of the
//             BatchNorm-backward sums (m1, m2) of the layer below: those are linear in Gp);
//   channels: dWs[c, :] += T_c A[winner row of c, :]  accumulated over the groups of a persistent workgroup in registers.
// Fixed orders everywhere (rows by ascending channel, groups by ascending index, partial blocks summed in split order): no
// atomics, the same bits from run to run.
#include "common.h"

namespace {

struct SparseArgs {
    int G, K;
    const int32_t *arg;                       // [G, Cout] winning row inside the group
    const float *T;                           // [G, Cout]
    const float *W;                           // [Cout, Cin] row-major
    const float *Yp; long long ldyp;          // [G K, Cin]
    const float *ps, *pt, *pmu, *pis;         // the layer below: scale, shift, mean, invstd [Cin]
    float *Gp; long long ldgp;                // [G K, Cin]: read-modify-written on the winners' rows
    float *red_slab;                          // [row grid][2][Cin]
    float *dws_part;                          // [channel splits][Cout][Cin]
    int nsplit;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Two launches (round 5, late): the first form did rows and channels in ONE persistent kernel of 256 workgroups -- a group
// after the other per workgroup, two barriers per group, every load at the end of a dependent chain: 174 us for 3072 groups
// of 64 rows (SA2 scale 1), 1254 us for 12288 groups of 128 (SA1 scale 3), more than the dense pass saved.  Rows and
// channels share nothing but their inputs:
//   rows     one workgroup per group (a grid-stride loop when there are more than SPARSE_ROW_GRID groups): (arg, T) of the group
//            in LDS, 8 waves, a wave per row, the channels that won a row from ballots in ascending order (8 rows of W in flight); the row's sum of T_c W[c, :]
//            is added to Gp[row] and enters the (m1, m2) sums of the layer below, one slab per workgroup.
//   channels one wave per channel and SPLIT of the groups: T_c A[winner row] over its groups in ascending order, 8 row loads in
//            flight (the (arg, T) pairs of 64 groups sit in the lanes); the per-split partial [Cout, Cin] blocks are added
//            by slab_sum_kernel in split order.
constexpr int SPARSE_ROW_GRID = 2048;

#ifndef PA_NW
#define PA_NW 4
#endif
#ifndef PA_UNR
#define PA_UNR 2
#endif
template <int COUT, int CIN>
__global__ __launch_bounds__(64 * PA_NW) void pool_alg_rows_kernel(const SparseArgs a)
{
    constexpr int NW = PA_NW, NTH = 64 * NW;
    constexpr int J = (CIN + 63) / 64;                       // columns per lane
    constexpr int NCH = COUT / 64;
    constexpr int UNR = PA_UNR;                              // W rows in flight per wave
    __shared__ int s_arg[COUT];
    __shared__ float s_T[COUT];
    __shared__ float s_red[NW][2][CIN];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float cps[J], cpt[J], cmu[J], cis[J], m1[J], m2[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int col = lane + 64 * j, cc = col < CIN ? col : 0;
        cps[j] = a.ps[cc]; cpt[j] = a.pt[cc]; cmu[j] = a.pmu[cc]; cis[j] = a.pis[cc];
        m1[j] = 0.f; m2[j] = 0.f;
    }
    for (int g = blockIdx.x; g < a.G; g += gridDim.x) {
        __syncthreads();                                     // (the previous group's rows have read s_arg / s_T)
        for (int t = threadIdx.x; t < COUT; t += NTH) {
            s_arg[t] = a.arg[(size_t)g * COUT + t];
            s_T[t] = a.T[(size_t)g * COUT + t];
        }
        __syncthreads();
        int myarg[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) myarg[q] = (s_T[64 * q + lane] != 0.f) ? s_arg[64 * q + lane] : -1;
        for (int r = wave; r < a.K; r += NW) {
            unsigned long long mk[NCH];
            bool any = false;
#pragma unroll
            for (int q = 0; q < NCH; ++q) { mk[q] = __ballot(myarg[q] == r); any = any || mk[q] != 0ull; }
            if (!any) continue;                              // (wave-uniform)
            const size_t row = (size_t)g * a.K + r;
            float y[J], old[J];                              // the row's own loads first: they fly under the channel walk
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int col = lane + 64 * j;
                y[j] = col < CIN ? a.Yp[row * a.ldyp + col] : 0.f;
                old[j] = col < CIN ? a.Gp[row * a.ldgp + col] : 0.f;
            }
            float acc[J];
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] = 0.f;
            // the channels that won this row in ascending order, UNR rows of W in flight at a time (one after the other each
            // costs an L2 round trip: a row of a 64-row group wins ~4 of 256 channels); the masks are wave-uniform, so the walk
            // over their bits is scalar work; a dead slot multiplies W's first row by zero
            while (any) {
                int cs[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    int c = -1;
#pragma unroll
                    for (int q = 0; q < NCH; ++q)
                        if (c < 0 && mk[q]) { c = 64 * q + __builtin_ctzll(mk[q]); mk[q] &= mk[q] - 1ull; }
                    cs[u] = c;
                }
                float w[UNR][J], ts[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int cc = cs[u] >= 0 ? cs[u] : 0;
                    ts[u] = cs[u] >= 0 ? s_T[cc] : 0.f;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int col = lane + 64 * j;
                        w[u][j] = col < CIN ? a.W[(size_t)cc * CIN + col] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int j = 0; j < J; ++j) acc[j] = fmaf(ts[u], w[u][j], acc[j]);
                any = false;
#pragma unroll
                for (int q = 0; q < NCH; ++q) any = any || mk[q] != 0ull;
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int col = lane + 64 * j;
                if (col < CIN) {
                    a.Gp[row * a.ldgp + col] = old[j] + acc[j];
                    const float gm = fmaf(y[j], cps[j], cpt[j]) > 0.f ? acc[j] : 0.f;
                    m1[j] += gm;
                    m2[j] += gm * ((y[j] - cmu[j]) * cis[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int col = lane + 64 * j;
        if (col < CIN) { s_red[wave][0][col] = m1[j]; s_red[wave][1][col] = m2[j]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * CIN; t += NTH) {
        const int which = t / CIN, col = t - which * CIN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += s_red[w][which][col];
        a.red_slab[((size_t)blockIdx.x * 2 + which) * CIN + col] = v;
    }
}

// grid (Cout / 4, nsplit), 256 threads: wave = channel, blockIdx.y = split of the groups
template <int COUT, int CIN>
__global__ __launch_bounds__(256) void pool_alg_channels_kernel(const SparseArgs a)
{
    constexpr int J = (CIN + 63) / 64;
    constexpr int UNR = 8;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = (a.G + a.nsplit - 1) / a.nsplit;
    const int g0 = blockIdx.y * per, g1 = min(a.G, g0 + per);
    float cps[J], cpt[J], acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int col = lane + 64 * j, cc = col < CIN ? col : 0;
        cps[j] = a.ps[cc]; cpt[j] = a.pt[cc]; acc[j] = 0.f;
    }
    for (int gc = g0; gc < g1; gc += 64) {                   // the (arg, T) pairs of 64 groups in the lanes
        const int cnt = min(64, g1 - gc);
        const size_t at = (size_t)(gc + (lane < cnt ? lane : 0)) * COUT + c;
        const int al = a.arg[at];
        const float tl = lane < cnt ? a.T[at] : 0.f;
        for (int u0 = 0; u0 < cnt; u0 += UNR) {
            float x[UNR][J], tt[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool live = u0 + u < cnt;              // (wave-uniform)
                const int ar = live ? __builtin_amdgcn_readlane(al, u0 + u) : 0;
                tt[u] = live ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tl), u0 + u)) : 0.f;
                // unconditional loads (a dead slot re-reads the chunk's first row): a branch around a load makes the compiler
                // wait for it at the join and the row loads of a batch would go out one by one
                const float *src = a.Yp + ((size_t)(gc + (live ? u0 + u : 0)) * a.K + ar) * a.ldyp + lane;
#pragma unroll
                for (int j = 0; j < J; ++j) x[u][j] = (lane + 64 * j) < CIN ? src[64 * j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = fmaf(tt[u], fmaxf(fmaf(x[u][j], cps[j], cpt[j]), 0.f), acc[j]);
        }
    }
    float *dst = a.dws_part + ((size_t)blockIdx.y * COUT + c) * CIN + lane;
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (lane + 64 * j < CIN) dst[64 * j] = acc[j];
}

// out[i] = sum over the slabs of part[slab][i], four interleaved chains in a fixed order
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ part, int nslab, long long n, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int w = 0;
    for (; w + 3 < nslab; w += 4) {
        a0 += part[(size_t)w * n + i]; a1 += part[(size_t)(w + 1) * n + i];
        a2 += part[(size_t)(w + 2) * n + i]; a3 += part[(size_t)(w + 3) * n + i];
    }
    for (; w < nslab; ++w) a0 += part[(size_t)w * n + i];
    out[i] = (a0 + a1) + (a2 + a3);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same S terms for a layer pooled over the WHOLE cloud (src/dgcnn.py:194-197: relu(gn(mlp1(.))) then max over the N points;
// one group per sample, K = N rows, Cout winners per sample): the dense terms are plain batched products on the host side
// (src/dgcnn.py of this package: per-sample M_b = W^T diag(b_b) W and Gram matrices), here only
//   rows:     dX[b, winner row] += sum over the channels c that row won of T[b, c] W[c, :]      (ascending c)
//   channels: dW[c, :]          += sum over the samples b of T[b, c] X[b, winner row of (b, c), :]   (ascending b)
// -- B x Cout = 24 576 winners for the whole batch, against 25.8 GFLOP per dense product over the [B N, Cout] tensor dY that
// then never exists.
// ---------------------------------------------------------------------------------------------------------------------------
template <int J>
__global__ __launch_bounds__(256) void global_pool_rows_kernel(const int32_t *__restrict__ arg, const float *__restrict__ T,
                                                               const float *__restrict__ W, long long ldw, int K, int Cout,
                                                               int rows_per_wg, float *__restrict__ dX, long long ldd)
{
    constexpr int MAXQ = 16;                                   // Cout <= 1024
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = Cout >> 6;
    int myarg[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        myarg[q] = -1;
        if (q < nq) {
            const size_t at = (size_t)b * Cout + 64 * q + lane;
            myarg[q] = T[at] != 0.f ? arg[at] : -1;
        }
    }
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(K, r0 + rows_per_wg);
    for (int r = r0 + wave; r < r1; r += 4) {
        unsigned long long mk[MAXQ];
        bool any = false;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            mk[q] = q < nq ? __ballot(myarg[q] == r) : 0ull;
            any = any || mk[q] != 0ull;
        }
        if (!any) continue;                                    // (wave-uniform)
        float acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = 0.f;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            unsigned long long m = mk[q];
            while (m) {
                const int c = 64 * q + __builtin_ctzll(m);
                m &= m - 1ull;
                const float t = T[(size_t)b * Cout + c];
#pragma unroll
                for (int j = 0; j < J; ++j) acc[j] = fmaf(t, W[(size_t)c * ldw + lane + 64 * j], acc[j]);
            }
        }
        float *dst = dX + ((size_t)b * K + r) * ldd + lane;
#pragma unroll
        for (int j = 0; j < J; ++j) dst[64 * j] += acc[j];
    }
}

template <int J>
__global__ __launch_bounds__(256) void global_pool_channels_kernel(const int32_t *__restrict__ arg, const float *__restrict__ T,
                                                                   const float *__restrict__ X, long long ldx, int Bs, int K,
                                                                   int Cout, float *__restrict__ dW, long long lddw)
{
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= Cout) return;
    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.f;
    for (int b = 0; b < Bs; ++b) {
        const float t = T[(size_t)b * Cout + c];
        if (t == 0.f) continue;                                // (wave-uniform)
        const float *src = X + ((size_t)b * K + arg[(size_t)b * Cout + c]) * ldx + lane;
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = fmaf(t, src[64 * j], acc[j]);
    }
    float *dst = dW + (size_t)c * lddw + lane;
#pragma unroll
    for (int j = 0; j < J; ++j) dst[64 * j] += acc[j];
}

int sparse_grid(int G) { return G < SPARSE_ROW_GRID ? G : SPARSE_ROW_GRID; }

// splits of the groups in the channel pass: ~2048 workgroups in all, at least 64 groups per split
int sparse_splits(int G, int Cout)
{
    int ns = 2048 / (Cout / 4);
    if (ns > (G + 63) / 64) ns = (G + 63) / 64;
    return ns < 1 ? 1 : ns;
}

bool sparse_shape_ok(int Cout, int Cin)
{
    return (Cout == 128 && (Cin == 96 || Cin == 64)) || (Cout == 256 && Cin == 128) || (Cout == 64 && Cin == 32);
}

}  // namespace

extern "C" {

int prifit_pool_alg_sparse_supported(int G, int K, int Cout, int Cin) { return (G > 0 && K > 0 && sparse_shape_ok(Cout, Cin)) ? 1 : 0; }
int prifit_pool_alg_sparse_slabs(int G) { return G > 0 ? sparse_grid(G) : 0; }
long long prifit_pool_alg_sparse_workspace(int G, int Cout, int Cin)
{
    return (G > 0 && sparse_shape_ok(Cout, Cin)) ? (long long)sparse_splits(G, Cout) * Cout * Cin : 0;
}

int prifit_pool_alg_sparse_f32(int G, int K, int Cout, int Cin, const int32_t *arg, const float *T, const float *W,
                               const float *Yp, long long ldyp, const float *p_scale, const float *p_shift, const float *p_mean,
                               const float *p_invstd, float *Gp, long long ldgp, float *red_slab, float *dWs, float *workspace,
                               void *stream)
{
    if (!arg || !T || !W || !Yp || !p_scale || !p_shift || !p_mean || !p_invstd || !Gp || !red_slab || !dWs || !workspace ||
        !prifit_pool_alg_sparse_supported(G, K, Cout, Cin) || ldyp < Cin || ldgp < Cin)
        return PRIFIT_EINVAL;
    SparseArgs a;
    a.G = G; a.K = K; a.arg = arg; a.T = T; a.W = W; a.Yp = Yp; a.ldyp = ldyp; a.ps = p_scale; a.pt = p_shift; a.pmu = p_mean;
    a.pis = p_invstd; a.Gp = Gp; a.ldgp = ldgp; a.red_slab = red_slab; a.dws_part = workspace;
    a.nsplit = sparse_splits(G, Cout);
    const dim3 rgrid((unsigned)sparse_grid(G)), cgrid((unsigned)(Cout / 4), (unsigned)a.nsplit);
    hipStream_t st = as_stream(stream);
#define SPARSE(CO, CI)                                                                          \
    do {                                                                                        \
        hipLaunchKernelGGL((pool_alg_rows_kernel<CO, CI>), rgrid, dim3(64 * PA_NW), 0, st, a);         \
        hipLaunchKernelGGL((pool_alg_channels_kernel<CO, CI>), cgrid, dim3(256), 0, st, a);     \
    } while (0)
    if (Cout == 128 && Cin == 96) SPARSE(128, 96);
    else if (Cout == 128 && Cin == 64) SPARSE(128, 64);
    else if (Cout == 256 && Cin == 128) SPARSE(256, 128);
    else SPARSE(64, 32);
#undef SPARSE
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, a.nsplit, n, dWs);
    return prifit_check_launch();
}

int prifit_global_pool_winners_supported(int Cout, int Cin)
{
    return (Cout > 0 && Cout % 64 == 0 && Cout <= 1024 && (Cin == 64 || Cin == 128 || Cin == 256)) ? 1 : 0;
}

int prifit_global_pool_winners_f32(int Bs, int K, int Cout, int Cin, const int32_t *arg, const float *T, const float *W,
                                   long long ldw, const float *X, long long ldx, float *dX, long long lddx, float *dW,
                                   long long lddw, void *stream)
{
    if (!arg || !T || Bs <= 0 || Bs > 65535 || K <= 0 || !prifit_global_pool_winners_supported(Cout, Cin) || (dX && (!W || ldw < Cin || lddx < Cin)) ||
        (dW && (!X || ldx < Cin || lddw < Cin)))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if (dX) {
        const int rows_per_wg = 64;
        const dim3 grid((unsigned)((K + rows_per_wg - 1) / rows_per_wg), (unsigned)Bs);
        if (Cin == 256) hipLaunchKernelGGL(global_pool_rows_kernel<4>, grid, dim3(256), 0, st, arg, T, W, ldw, K, Cout, rows_per_wg, dX, lddx);
        else if (Cin == 128) hipLaunchKernelGGL(global_pool_rows_kernel<2>, grid, dim3(256), 0, st, arg, T, W, ldw, K, Cout, rows_per_wg, dX, lddx);
        else hipLaunchKernelGGL(global_pool_rows_kernel<1>, grid, dim3(256), 0, st, arg, T, W, ldw, K, Cout, rows_per_wg, dX, lddx);
    }
    if (dW) {
        const dim3 grid((unsigned)((Cout + 3) / 4));
        if (Cin == 256) hipLaunchKernelGGL(global_pool_channels_kernel<4>, grid, dim3(256), 0, st, arg, T, X, ldx, Bs, K, Cout, dW, lddw);
        else if (Cin == 128) hipLaunchKernelGGL(global_pool_channels_kernel<2>, grid, dim3(256), 0, st, arg, T, X, ldx, Bs, K, Cout, dW, lddw);
        else hipLaunchKernelGGL(global_pool_channels_kernel<1>, grid, dim3(256), 0, st, arg, T, X, ldx, Bs, K, Cout, dW, lddw);
    }
    return prifit_check_launch();
}

int prifit_slab_sum(const float *part, int nslab, long long n, float *out, void *stream)
{
    if (!part || !out || nslab <= 0 || n <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), part, nslab, n, out);
    return prifit_check_launch();
}

}  // extern "C"
