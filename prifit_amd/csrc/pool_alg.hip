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
// Fixed orders everywhere (rows by ascending channel, groups by ascending index, partial slabs summed by a second launch): no
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
    float *red_slab;                          // [grid][2][Cin]
    float *dws_part;                          // [grid][Cout][Cin]
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int COUT, int CIN, int NTH>
__global__ __launch_bounds__(NTH) void pool_alg_sparse_kernel(const SparseArgs a)
{
    constexpr int NW = NTH / 64;
    constexpr int PARTS = NTH / COUT, CPT = CIN / PARTS;     // threads per channel, columns per thread (channel phase)
    constexpr int J = (CIN + 63) / 64;                       // columns per lane (row phase)
    constexpr int NCH = COUT / 64;
    static_assert(NTH % COUT == 0 && CIN % PARTS == 0 && CPT % 4 == 0 && COUT % 64 == 0, "thread mapping");
    __shared__ int s_arg[COUT];
    __shared__ float s_T[COUT];
    __shared__ __attribute__((aligned(16))) float s_ps[CIN], s_pt[CIN];
    __shared__ float s_red[NW][2][CIN];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int t = threadIdx.x; t < CIN; t += NTH) { s_ps[t] = a.ps[t]; s_pt[t] = a.pt[t]; }
    // row phase: this lane's columns and their BatchNorm constants
    float cps[J], cpt[J], cmu[J], cis[J], m1[J], m2[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int col = lane + 64 * j, cc = col < CIN ? col : 0;
        cps[j] = a.ps[cc]; cpt[j] = a.pt[cc]; cmu[j] = a.pmu[cc]; cis[j] = a.pis[cc];
        m1[j] = 0.f; m2[j] = 0.f;
    }
    // channel phase: this thread's channel and column range, its accumulators
    const int ch = threadIdx.x % COUT, part = threadIdx.x / COUT;
    float4 accw[CPT / 4];
#pragma unroll
    for (int i = 0; i < CPT / 4; ++i) accw[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int g = blockIdx.x; g < a.G; g += gridDim.x) {
        __syncthreads();                                     // (the previous group's channel phase has read s_arg / s_T)
        for (int t = threadIdx.x; t < COUT; t += NTH) {
            s_arg[t] = a.arg[(size_t)g * COUT + t];
            s_T[t] = a.T[(size_t)g * COUT + t];
        }
        __syncthreads();
        // ---- rows: wave w takes rows w, w + NW, ...; the channels that won a row come out of ballots, ascending
        int myarg[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) myarg[q] = (s_T[64 * q + lane] != 0.f) ? s_arg[64 * q + lane] : -1;
        for (int r = wave; r < a.K; r += NW) {
            unsigned long long mk[NCH];
            bool any = false;
#pragma unroll
            for (int q = 0; q < NCH; ++q) { mk[q] = __ballot(myarg[q] == r); any = any || mk[q] != 0ull; }
            if (!any) continue;                              // (wave-uniform)
            float acc[J];
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] = 0.f;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                unsigned long long m = mk[q];
                while (m) {
                    const int c = 64 * q + __builtin_ctzll(m);
                    m &= m - 1ull;
                    const float t = s_T[c];
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int col = lane + 64 * j;
                        if (col < CIN) acc[j] = fmaf(t, a.W[(size_t)c * CIN + col], acc[j]);
                    }
                }
            }
            const size_t row = (size_t)g * a.K + r;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int col = lane + 64 * j;
                if (col < CIN) {
                    const float y = a.Yp[row * a.ldyp + col];
                    float *dst = a.Gp + row * a.ldgp + col;
                    *dst += acc[j];
                    const float gm = fmaf(y, cps[j], cpt[j]) > 0.f ? acc[j] : 0.f;
                    m1[j] += gm;
                    m2[j] += gm * ((y - cmu[j]) * cis[j]);
                }
            }
        }
        // ---- channels: thread (channel, column range)
        {
            const float t = s_T[ch];
            if (t != 0.f) {
                const float *src = a.Yp + ((size_t)g * a.K + s_arg[ch]) * a.ldyp + part * CPT;
#pragma unroll
                for (int i = 0; i < CPT / 4; ++i) {
                    const float4 y = ld4(src + 4 * i);
                    const float4 s = *reinterpret_cast<const float4 *>(&s_ps[part * CPT + 4 * i]);
                    const float4 sh = *reinterpret_cast<const float4 *>(&s_pt[part * CPT + 4 * i]);
                    accw[i].x = fmaf(t, fmaxf(fmaf(y.x, s.x, sh.x), 0.f), accw[i].x);
                    accw[i].y = fmaf(t, fmaxf(fmaf(y.y, s.y, sh.y), 0.f), accw[i].y);
                    accw[i].z = fmaf(t, fmaxf(fmaf(y.z, s.z, sh.z), 0.f), accw[i].z);
                    accw[i].w = fmaf(t, fmaxf(fmaf(y.w, s.w, sh.w), 0.f), accw[i].w);
                }
            }
        }
    }
    // ---- this workgroup's partials
    float *dst = a.dws_part + ((size_t)blockIdx.x * COUT + ch) * CIN + part * CPT;
#pragma unroll
    for (int i = 0; i < CPT / 4; ++i) *reinterpret_cast<float4 *>(dst + 4 * i) = accw[i];
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

int sparse_grid(int G) { return G < 256 ? G : 256; }

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
    return (G > 0 && sparse_shape_ok(Cout, Cin)) ? (long long)sparse_grid(G) * Cout * Cin : 0;
}

int prifit_pool_alg_sparse_f32(int G, int K, int Cout, int Cin, const int32_t *arg, const float *T, const float *W,
                               const float *Yp, long long ldyp, const float *p_scale, const float *p_shift, const float *p_mean,
                               const float *p_invstd, float *Gp, long long ldgp, float *red_slab, float *dWs, float *workspace,
                               void *stream)
{
    if (!arg || !T || !W || !Yp || !p_scale || !p_shift || !p_mean || !p_invstd || !Gp || !red_slab || !dWs || !workspace ||
        !prifit_pool_alg_sparse_supported(G, K, Cout, Cin) || ldyp < Cin || (ldyp & 3) || ldgp < Cin ||
        (((uintptr_t)Yp | (uintptr_t)workspace) & 15))
        return PRIFIT_EINVAL;
    SparseArgs a;
    a.G = G; a.K = K; a.arg = arg; a.T = T; a.W = W; a.Yp = Yp; a.ldyp = ldyp; a.ps = p_scale; a.pt = p_shift; a.pmu = p_mean;
    a.pis = p_invstd; a.Gp = Gp; a.ldgp = ldgp; a.red_slab = red_slab; a.dws_part = workspace;
    const int grid = sparse_grid(G);
    hipStream_t st = as_stream(stream);
    if (Cout == 128 && Cin == 96) hipLaunchKernelGGL((pool_alg_sparse_kernel<128, 96, 256>), dim3(grid), dim3(256), 0, st, a);
    else if (Cout == 128 && Cin == 64) hipLaunchKernelGGL((pool_alg_sparse_kernel<128, 64, 256>), dim3(grid), dim3(256), 0, st, a);
    else if (Cout == 256 && Cin == 128) hipLaunchKernelGGL((pool_alg_sparse_kernel<256, 128, 512>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((pool_alg_sparse_kernel<64, 32, 256>), dim3(grid), dim3(256), 0, st, a);
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, grid, n, dWs);
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
