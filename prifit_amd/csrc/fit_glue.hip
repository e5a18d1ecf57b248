// Small per-shape reductions of the fitting path, one launch each, in place of chains of elementwise / reduce launches of
// a few dozen elements (profiles/r03_c3_launch_census.txt: ~55 launches and 0.25 ms of a 15.5 ms step, and as many
// host-side dispatches).  Nothing here is bandwidth- or matrix-bound; the point is the launch count.
#include "common.h"

// bw[b] = mean_i sqrt(max(kth[b][i], 1e-6))   (src/mean_shift.py:156-160 after the k-th smallest chord distance per row)
__global__ __launch_bounds__(256) void bandwidth_from_kth_kernel(const float *__restrict__ kth, int N, float *__restrict__ bw)
{
    __shared__ double s_part[4];
    const int b = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) acc += (double)sqrtf(fmaxf(kth[(size_t)b * N + i], 1e-6f));
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bw[b] = (float)((s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (double)N);
}

// guard_mean_shift's check (src/ellipsoid_utils.py:19-27) for all shapes: bad = any shape whose distinct-label count exceeds
// max_clusters (or keeps more than `slots` centres).  nuniq[b] = count[b] when nms kept more than `cap` centres (labels then
// cover only the first cap), else the number of used label slots.
__global__ __launch_bounds__(256) void cluster_verdict_kernel(const int32_t *__restrict__ count,
                                                              const int32_t *__restrict__ used, int B, int cap,
                                                              int max_clusters, int slots, int32_t *__restrict__ nuniq,
                                                              int32_t *__restrict__ bad)
{
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < B; b += 4) {
        int n = 0;
        for (int k = lane; k < cap; k += 64) n += used[(size_t)b * cap + k] != 0 ? 1 : 0;
        n = wave_sum_i32_dpp(n);
        const int c = count[b];
        const int u = c > cap ? c : n;
        if (lane == 0) {
            if (nuniq) nuniq[b] = u;
            if (u > max_clusters || c > slots) atomicOr(&s_bad, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) bad[0] = s_bad;
}

// gmax[b] = (max over points j and live clusters k < count[b] of dots[b][j][k]) / bw[b]^2  (src/mean_shift.py:237-242: the
// GLOBAL maximum of the similarity matrix, detached); -inf / bw^2 when a shape has no cluster (as the masked amax gives).
// Two launches: GM_PARTS workgroups per shape reduce a slice each (16-byte loads), one wave per shape combines.
constexpr int GM_PARTS = 16;
__global__ __launch_bounds__(256) void membership_gmax_part_kernel(const float *__restrict__ dots, const int32_t *__restrict__ count,
                                                                   int N, int KM, float *__restrict__ part)
{
    __shared__ float s_part[4];
    const int b = blockIdx.y;
    const int K = min(count[b], KM);
    const float4 *d = reinterpret_cast<const float4 *>(dots + (size_t)b * N * KM);
    const int q = KM >> 2;                               // float4 per row (KM % 4 == 0)
    const long long total = (long long)N * q;
    const long long per = (total + GM_PARTS - 1) / GM_PARTS;
    const long long lo = blockIdx.x * per, hi = min(total, lo + per);
    float m = -INFINITY;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const int k = (int)(i % q) * 4;
        const float4 v = d[i];
        m = (k + 0 < K && v.x > m) ? v.x : m;
        m = (k + 1 < K && v.y > m) ? v.y : m;
        m = (k + 2 < K && v.z > m) ? v.z : m;
        m = (k + 3 < K && v.w > m) ? v.w : m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[b * GM_PARTS + blockIdx.x] = fmaxf(fmaxf(s_part[0], s_part[1]), fmaxf(s_part[2], s_part[3]));
}

__global__ __launch_bounds__(64) void membership_gmax_final_kernel(const float *__restrict__ part, const float *__restrict__ bw,
                                                                   int B, float *__restrict__ gmax)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < GM_PARTS; ++i) m = fmaxf(m, part[b * GM_PARTS + i]);
    gmax[b] = m / (bw[b] * bw[b]);
}

// analytic_chamfer_distance's combination (src/utils.py:417-426): per shape (d2_sum / max(total, 1) + sdf_sum / M) / 2,
// averaged over the shapes that have at least one primitive (0 when none has).  part[0][b], part[1][b]: the two halves.
__global__ __launch_bounds__(64) void chamfer_combine_fwd_kernel(const float *__restrict__ d2_sum, const int32_t *__restrict__ total,
                                                                 const float *__restrict__ sdf_sum, const int32_t *__restrict__ valid,
                                                                 int B, int KM, int M, float *__restrict__ loss,
                                                                 float *__restrict__ part, float *__restrict__ coef)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    int nh = 0;
    for (int b = lane; b < B; b += 64) {
        int has = 0;
        for (int k = 0; k < KM; ++k) has |= valid[(size_t)b * KM + k] != 0 ? 1 : 0;
        const float t = (float)max(total[b], 1);
        const float pd = d2_sum[b] / t, ps = sdf_sum[b] / (float)M;
        part[b] = pd;
        part[B + b] = ps;
        acc += has ? (pd + ps) / 2.0f : 0.f;
        nh += has;
        coef[b] = has ? 1.0f / t : 0.f;        // d loss / d d2_sum[b] up to the common factor 1 / (2 nh)
        coef[B + b] = has ? 1.0f / (float)M : 0.f;
    }
    acc = wave_sum_f32(acc);
    nh = wave_sum_i32_dpp(nh);
    const float denom = (float)max(nh, 1);
    if (lane == 0) {
        loss[0] = acc / denom;
        coef[2 * B] = 0.5f / denom;
    }
}

__global__ __launch_bounds__(64) void chamfer_combine_bwd_kernel(const float *__restrict__ g, const float *__restrict__ coef, int B,
                                                                 float *__restrict__ g_d2, float *__restrict__ g_sdf)
{
    const float s = g[0] * coef[2 * B];
    for (int b = threadIdx.x; b < B; b += 64) {
        g_d2[b] = s * coef[b];
        g_sdf[b] = s * coef[B + b];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The segmentation loss (models/pointnet2_part_seg_msg.py:137-144: F.cross_entropy(pred, target), mean over the B N points, on
// rows of C <= 64 class scores): loss = mean_r (lse_r - x[r, target_r]), lse_r = max + log sum exp(x - max).
// torch's nll_loss_forward / _backward reduce with ONE workgroup (73 + 47 us for 49152 x 50 at B = 24); here a wave per row,
// per-workgroup partial sums in a fixed row order, a second tiny launch for the mean.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CE_WGS = 512;
constexpr long long CE_IGNORE = -100;    // F.cross_entropy's default ignore_index

__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float *__restrict__ x, long long ld, const long long *__restrict__ target,
                                                     long long P, int C, float *__restrict__ lse, float *__restrict__ part)
{
    __shared__ float s_w[4], s_n[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f, kept = 0.f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < P; r += (long long)gridDim.x * 4) {
        const float v = lane < C ? x[r * ld + lane] : -INFINITY;
        const float m = wave_max_f32(v);
        const float e = lane < C ? __expf(v - m) : 0.f;
        const float l = m + __logf(wave_sum_f32(e));
        const long long t = target[r];
        if (lane == 0) lse[r] = l;
        if (t == CE_IGNORE) continue;                    // ignore_index = -100: no term, not counted (wave-uniform)
        // any other label outside [0, C) is an error: torch stops the process with a device-side assert, here the loss (and,
        // in the backward, the row's gradient) turns into NaN -- loud, and without a host read-back
        const float xt = (t < 0 || t >= C) ? __builtin_nanf("") : __shfl(v, (int)t, 64);
        acc += l - xt;                                   // (the same value in every lane)
        kept += 1.f;
    }
    if (lane == 0) { s_w[wave] = acc; s_n[wave] = kept; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        part[CE_WGS + blockIdx.x] = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
    }
}

// out[0] = mean over the KEPT rows (0 / 0 = NaN when every row is ignored, as torch), out[1] = the number of kept rows
__global__ __launch_bounds__(64) void ce_final_kernel(const float *__restrict__ part, int n, float *__restrict__ out)
{
    double a = 0.0, k = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) { a += (double)part[i]; k += (double)part[CE_WGS + i]; }
    a = wave_sum_f64(a);
    k = wave_sum_f64(k);
    if (threadIdx.x == 0) { out[0] = (float)(a / k); out[1] = (float)k; }
}

// d loss / d x[r, c] = (softmax(x)[r, c] - [c == target_r]) g / kept; ignored rows get zeros, rows with an invalid label NaN
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float *__restrict__ x, long long ld, const long long *__restrict__ target,
                                                     const float *__restrict__ lse, const float *__restrict__ g,
                                                     const float *__restrict__ kept, long long P, int C, float *__restrict__ dx,
                                                     long long ldd)
{
    const long long total = P * C;
    const float gs = g[0] / kept[0];
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long r = id / C;
        const int c = (int)(id - r * C);
        const long long t = target[r];
        float v;
        if (t == CE_IGNORE) v = 0.f;
        else if (t < 0 || t >= C) v = __builtin_nanf("");
        else v = (__expf(x[r * ld + c] - lse[r]) - (c == (int)t ? 1.f : 0.f)) * gs;
        dx[r * ldd + c] = v;
    }
}

extern "C" {

int prifit_bandwidth_from_kth(const float *kth, int B, int N, float *bw, void *stream)
{
    if (!kth || !bw || B <= 0 || N <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(bandwidth_from_kth_kernel, dim3(B), dim3(256), 0, as_stream(stream), kth, N, bw);
    return prifit_check_launch();
}

int prifit_cluster_verdict(const int32_t *count, const int32_t *used, int B, int cap, int max_clusters, int slots,
                           int32_t *nuniq, int32_t *bad, void *stream)
{
    if (!count || !used || !bad || B <= 0 || cap <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(cluster_verdict_kernel, dim3(1), dim3(256), 0, as_stream(stream), count, used, B, cap, max_clusters,
                       slots, nuniq, bad);
    return prifit_check_launch();
}

long long prifit_membership_gmax_workspace(int B) { return (long long)B * GM_PARTS; }

int prifit_membership_gmax(const float *dots, const float *bw, const int32_t *count, int B, int N, int KM, float *gmax,
                           float *workspace, void *stream)
{
    if (!dots || !bw || !count || !gmax || !workspace || B <= 0 || N <= 0 || KM <= 0 || (KM & 3) || ((uintptr_t)dots & 15))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(membership_gmax_part_kernel, dim3(GM_PARTS, B), dim3(256), 0, as_stream(stream), dots, count, N, KM,
                       workspace);
    hipLaunchKernelGGL(membership_gmax_final_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), workspace, bw, B,
                       gmax);
    return prifit_check_launch();
}

int prifit_chamfer_combine_fwd(const float *d2_sum, const int32_t *total, const float *sdf_sum, const int32_t *valid, int B,
                               int KM, int M, float *loss, float *part, float *coef, void *stream)
{
    if (!d2_sum || !total || !sdf_sum || !valid || !loss || !part || !coef || B <= 0 || KM <= 0 || M <= 0)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(chamfer_combine_fwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), d2_sum, total, sdf_sum, valid, B,
                       KM, M, loss, part, coef);
    return prifit_check_launch();
}

int prifit_chamfer_combine_bwd(const float *g, const float *coef, int B, float *g_d2, float *g_sdf, void *stream)
{
    if (!g || !coef || !g_d2 || !g_sdf || B <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(chamfer_combine_bwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), g, coef, B, g_d2, g_sdf);
    return prifit_check_launch();
}

int prifit_cross_entropy_workspace(void) { return 2 * CE_WGS; }

int prifit_cross_entropy_fwd(const float *x, long long ld, const long long *target, long long P, int C, float *lse, float *workspace,
                             float *loss, void *stream)
{
    if (!x || !target || !lse || !workspace || !loss || P <= 0 || C <= 0 || C > 64 || ld < C) return PRIFIT_EINVAL;
    const int grid = (int)((P + 3) / 4 < CE_WGS ? (P + 3) / 4 : CE_WGS);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, ld, target, P, C, lse, workspace);
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), workspace, grid, loss);
    return prifit_check_launch();
}

int prifit_cross_entropy_bwd(const float *x, long long ld, const long long *target, const float *lse, const float *g,
                             const float *kept, long long P, int C, float *dx, long long ldd, void *stream)
{
    if (!x || !target || !lse || !g || !kept || !dx || P <= 0 || C <= 0 || C > 64 || ld < C || ldd < C) return PRIFIT_EINVAL;
    long long grid = (P * C + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, ld, target, lse, g, kept, P, C, dx, ldd);
    return prifit_check_launch();
}

}  // extern "C"
