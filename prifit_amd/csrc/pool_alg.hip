// A layer max-pooled over the WHOLE cloud (src/dgcnn.py:194-197), backward in the ALGEBRAIC form: the winners' terms.
//
// With dY = T [row == winner] + b Y + d and Y = A W^T + bias (A the layer's input, T the pooled gradient through GroupNorm +
// ReLU at each sample's winning row):
//     dA = A M + 1 v^T  +  S W                     S [P, Cout]: T at (winner row, channel), zero elsewhere
//     dW = diag(b) W (A^T A) + (d + b * bias) (1^T A)  +  S^T A
// The dense terms are per-sample [Cin, Cin] products on the host side (src/dgcnn.py of this package); this file adds the S terms as
// index work on the vector ALU, in fixed orders (no atomics, the same bits from run to run).
//
// Round 5 also built this form for the max-pooled set-abstraction layers (millions of winners per step): its dense pass was 30-55 %
// faster than the dA / dW pair it replaces, the winners' index work cost more than that saved (c2 9.98 -> 10.43 ms per step).
// Round 6 removed that arm (prifit_pool_alg_dense / _fused / _sparse); the measurements stay in DESIGN.md Appendix A and
// profiles/r05_ab_measurements.txt items 6 and 12.
#include "common.h"

namespace {
// out[i] = sum over the slabs of part[slab][i] in a fixed order.  Few outputs, many slabs (the first-layer weight gradient of a
// set-abstraction scale: 64 x 6 outputs from up to 1024 slabs): 32 outputs per workgroup, eight threads per output take every
// eighth slab in four interleaved chains, combined through LDS in thread order -- one thread per output walking all slabs was a
// serial chain of dependent adds on L2 latency in two workgroups (51 us per launch, three launches per step).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ part, int nslab, long long n, float *__restrict__ out)
{
    __shared__ float s_p[8][32];
    const int o = threadIdx.x & 31, l = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + o;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int w = l;
        for (; w + 24 < nslab; w += 32) {
            a0 += part[(size_t)w * n + i]; a1 += part[(size_t)(w + 8) * n + i];
            a2 += part[(size_t)(w + 16) * n + i]; a3 += part[(size_t)(w + 24) * n + i];
        }
        for (; w < nslab; w += 8) a0 += part[(size_t)w * n + i];
    }
    s_p[l][o] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (l == 0 && i < n) {
        float t = s_p[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += s_p[k][o];
        out[i] = t;
    }
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

}  // namespace

extern "C" {
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
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, as_stream(stream), part, nslab, n, out);
    return prifit_check_launch();
}

}  // extern "C"
