// Train-mode BatchNorm / ReLU / group max-pool kernels of the shared per-position MLP
// (models/pointnet_util.py:195-199, :252-256, :310-313).  All tensors are channels-last row-major
// matrices [P, ld] (P positions, C channels, C % 4 == 0, 16-byte aligned rows): every kernel moves
// float4s with consecutive lanes on consecutive channels, and per-channel reductions are done as
// per-block partial slabs [nblk][2][C] finalised in double precision (deterministic, no atomics).
#include "common.h"

constexpr int RED_ROWS = 128;  // rows of the matrix reduced by one workgroup

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4g(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// Column-slab reduction skeleton: threads are laid out as (C/4 float4 columns) x (row lanes).
// f(row, c4, acc0, acc1) accumulates two float4 partials per thread.
constexpr int POOL_RED_ROWS = 16;  // groups per workgroup in the pooled-layer reduction (only G = P/K rows: keep the grid wide)

template <int ROWS = RED_ROWS, typename F>
__device__ __forceinline__ void column_reduce(int P, int C, float *__restrict__ slab, F f, const BnTail &tail = BnTail{})
{
    __shared__ float4 s_red[2][256];
    __shared__ int s_tail;
    const int C4 = C >> 2;
    const int r_begin = blockIdx.x * ROWS, r_end = min(P, r_begin + ROWS);
    for (int cbase = 0; cbase < C4; cbase += 256) {
        const int cols = min(256, C4 - cbase);
        const int lanes = 256 / cols;  // row lanes
        const int c4 = cbase + threadIdx.x % cols, rl = threadIdx.x / cols;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (rl < lanes)
            for (int r = r_begin + rl; r < r_end; r += lanes) f(r, c4, a0, a1);
        s_red[0][threadIdx.x] = a0;
        s_red[1][threadIdx.x] = a1;
        __syncthreads();
        if (threadIdx.x < cols) {
            float4 t0 = s_red[0][threadIdx.x], t1 = s_red[1][threadIdx.x];
            for (int l = 1; l < lanes; ++l) {
                const float4 u0 = s_red[0][threadIdx.x + l * cols], u1 = s_red[1][threadIdx.x + l * cols];
                t0.x += u0.x; t0.y += u0.y; t0.z += u0.z; t0.w += u0.w;
                t1.x += u1.x; t1.y += u1.y; t1.z += u1.z; t1.w += u1.w;
            }
            if (tail.acc) {   // the sums finalized by this launch (common.h) instead of one slab per workgroup
                bn_tail_add(tail, 0, 4 * c4, t0.x); bn_tail_add(tail, 0, 4 * c4 + 1, t0.y);
                bn_tail_add(tail, 0, 4 * c4 + 2, t0.z); bn_tail_add(tail, 0, 4 * c4 + 3, t0.w);
                bn_tail_add(tail, 1, 4 * c4, t1.x); bn_tail_add(tail, 1, 4 * c4 + 1, t1.y);
                bn_tail_add(tail, 1, 4 * c4 + 2, t1.z); bn_tail_add(tail, 1, 4 * c4 + 3, t1.w);
            } else {
                st4g(slab + ((size_t)blockIdx.x * 2 + 0) * C + 4 * c4, t0);
                st4g(slab + ((size_t)blockIdx.x * 2 + 1) * C + 4 * c4, t1);
            }
        }
        __syncthreads();
    }
    if (tail.acc) bn_tail_finish(tail, &s_tail);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// Reduce the [nslab][2][C] partials -> batch mean / biased var -> scale/shift, running-stat update
// (torch.nn.BatchNorm semantics: running_var uses the unbiased estimate, momentum m).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ slab, int nslab, int C,
                                                          double count, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float momentum, float *running_mean,
                                                          float *running_var, float *__restrict__ scale,
                                                          float *__restrict__ shift, float *__restrict__ mean_o,
                                                          float *__restrict__ invstd_o)
{
    __shared__ double s_s[4], s_q[4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nslab; i += 256) {
        s += (double)slab[((size_t)i * 2 + 0) * C + c];
        q += (double)slab[((size_t)i * 2 + 1) * C + c];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if ((threadIdx.x & 63) == 0) { s_s[threadIdx.x >> 6] = s; s_q[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = s_s[0] + s_s[1] + s_s[2] + s_s[3];
        q = s_q[0] + s_q[1] + s_q[2] + s_q[3];
        float o[4];
        bn_fwd_coefs(s, q, count, gamma[c], beta[c], eps, momentum, running_mean ? running_mean + c : nullptr,
                     running_var ? running_var + c : nullptr, o, 1);
        scale[c] = o[0];
        shift[c] = o[1];
        mean_o[c] = o[2];
        invstd_o[c] = o[3];
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm (src/dgcnn.py:150-171: statistics per sample and channel group): the same finalize steps, one workgroup per
// (group, sample).  slab [Bs * sps][2][C]: `sps` consecutive slabs belong to one sample.  Replaces ~15 (forward) / ~20
// (backward) single-workgroup torch launches per layer, fp64 like them.
// ---------------------------------------------------------------------------------------------
// per-channel totals of both statistics of (sample b, the cpg channels from c0): thread t < cpg ends up with channel c0 + t
__device__ __forceinline__ void gn_channel_sums(const float *__restrict__ slab, int b, int sps, int C, int c0, int cpg,
                                                double (*s_part)[256], double &t0, double &t1)
{
    const int c = threadIdx.x % cpg, q = threadIdx.x / cpg, nq = 256 / cpg;
    double a0 = 0.0, a1 = 0.0;
    for (int i = q; i < sps; i += nq) {
        const float *p = slab + ((size_t)(b * sps + i) * 2) * C + c0 + c;
        a0 += (double)p[0];
        a1 += (double)p[C];
    }
    s_part[0][threadIdx.x] = a0;
    s_part[1][threadIdx.x] = a1;
    __syncthreads();
    t0 = 0.0; t1 = 0.0;
    if (threadIdx.x < cpg)
        for (int k = 0; k < nq; ++k) { t0 += s_part[0][k * cpg + threadIdx.x]; t1 += s_part[1][k * cpg + threadIdx.x]; }
    __syncthreads();
}

// sum over the first cpg threads' values (cpg <= 256), broadcast to all threads
__device__ __forceinline__ double gn_group_sum(double v, int cpg, double *s_red)
{
    v = threadIdx.x < cpg ? v : 0.0;
    v = wave_sum_f64(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return r;
}

// offset (may be NULL) [Bs][C]: the normalised tensor is Y + offset[b][c] (a per-sample, per-channel constant that the
// producer left out of Y, e.g. the part of a 1x1 convolution whose input is the same for every point of the sample); the
// slabs hold the statistics of Y alone, `rows` = rows per sample.  The tables come out RELATIVE TO Y -- mean' = mean -
// offset, shift' = beta - mean' scale -- so every consumer (affine / pool / backward kernels) works on Y unchanged.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ slab, int sps, int C, int cpg, double m,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          double eps, const float *__restrict__ offset, double rows,
                                                          float *__restrict__ scale, float *__restrict__ shift,
                                                          float *__restrict__ mean_o, float *__restrict__ invstd_o,
                                                          double *__restrict__ chsum)
{
    __shared__ double s_part[2][256];
    __shared__ double s_red[4];
    const int g = blockIdx.x, b = blockIdx.y, c0 = g * cpg;
    double t0, t1;
    gn_channel_sums(slab, b, sps, C, c0, cpg, s_part, t0, t1);
    if (chsum && threadIdx.x < cpg) chsum[(size_t)b * C + c0 + threadIdx.x] = t0;     // of Y alone (before the offset)
    double off = 0.0;
    if (offset && threadIdx.x < cpg) {
        off = (double)offset[(size_t)b * C + c0 + threadIdx.x];
        t1 += 2.0 * off * t0 + rows * off * off;     // sum (y + o)^2 = sum y^2 + 2 o sum y + n o^2
        t0 += rows * off;
    }
    const double s1 = gn_group_sum(t0, cpg, s_red) / m, s2 = gn_group_sum(t1, cpg, s_red) / m;
    double var = s2 - s1 * s1;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + eps));
    if (threadIdx.x < cpg) {
        const int c = c0 + threadIdx.x;
        const float mean = (float)(s1 - off);
        const float sc = __fmul_rn(gamma[c], invstd);
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = __fsub_rn(beta[c], __fmul_rn(mean, sc));
        mean_o[(size_t)b * C + c] = mean;
        invstd_o[(size_t)b * C + c] = invstd;
    }
}

// slab: (sum Gm, sum Gm * yhat) partials.  Out: cb, cd [Bs][C] (ca = scale), and the per-sample channel totals
// S [Bs][2][C] in fp64 (dgamma = sum_b S[b][1], dbeta = sum_b S[b][0]: two small reductions left to the caller).
// chsum (may be NULL) [Bs][C]: the forward's column sums of Y; then dsum [Bs][C] = the column sums of dY per sample WITHOUT
// reading dY: dY = ca Gm + cb Y + cd row by row, so sum_rows dY = ca sum Gm + cb sum Y + cd rows (the bias / offset gradient of
// the convolution in front: torch's reduction over the [B N, C] tensor took 16-32 us per layer).
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float *__restrict__ slab, int sps, int C, int cpg, double m,
                                                              const float *__restrict__ gamma, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, float *__restrict__ cb,
                                                              float *__restrict__ cd, double *__restrict__ S,
                                                              const double *__restrict__ chsum, double rows,
                                                              float *__restrict__ dsum)
{
    __shared__ double s_part[2][256];
    __shared__ double s_red[4];
    const int g = blockIdx.x, b = blockIdx.y, c0 = g * cpg;
    double t0, t1;
    gn_channel_sums(slab, b, sps, C, c0, cpg, s_part, t0, t1);
    const double gd = threadIdx.x < cpg ? (double)gamma[c0 + threadIdx.x] : 0.0;
    const double m1 = gn_group_sum(gd * t0, cpg, s_red) / m, m2 = gn_group_sum(gd * t1, cpg, s_red) / m;
    if (threadIdx.x < cpg) {
        const size_t o = (size_t)b * C + c0 + threadIdx.x;
        const double isd = (double)invstd[o], mu = (double)mean[o];
        const float cbf = (float)(-(isd * isd) * m2), cdf = (float)(-isd * m1 + mu * isd * isd * m2);
        cb[o] = cbf;
        cd[o] = cdf;
        if (chsum) {                                         // with the ROUNDED coefficients the apply pass multiplies by
            const float caf = __fmul_rn(gamma[c0 + threadIdx.x], invstd[o]);
            dsum[o] = (float)((double)caf * t0 + (double)cbf * chsum[o] + (double)cdf * rows);
        }
        S[((size_t)b * 2 + 0) * C + c0 + threadIdx.x] = t0;
        S[((size_t)b * 2 + 1) * C + c0 + threadIdx.x] = t1;
    }
}

// dbeta[c] = sum_b S[b][0][c], dgamma[c] = sum_b S[b][1][c] (fp64 sums of prifit_gn_bwd_finalize's per-sample pairs): one
// launch in place of torch's reduction + cast pair (~20 us for 2 C numbers, seven GroupNorm layers per step).
// dsum (may be NULL) [Bs][C]: db [C] = its sum over the samples (the bias gradient), threads 2 C .. 3 C
__global__ __launch_bounds__(256) void gn_param_grads_kernel(const double *__restrict__ S, int Bs, int C, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta, const float *__restrict__ dsum,
                                                             float *__restrict__ db)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * C) {
        if (dsum && t < 3 * C) {
            double acc = 0.0;
            for (int b = 0; b < Bs; ++b) acc += (double)dsum[(size_t)b * C + (t - 2 * C)];
            db[t - 2 * C] = (float)acc;
        }
        return;
    }
    double acc = 0.0;
    for (int b = 0; b < Bs; ++b) acc += S[(size_t)b * 2 * C + t];
    if (t < C) dbeta[t] = (float)acc;
    else dgamma[t - C] = (float)acc;
}

// out[c] += sum over the rows of Y[r][c]: the gradient of a convolution's bias (the decoders' and heads' biased 1x1 convolutions,
// src/dgcnn.py:236-259, models/pointnet2_part_seg_msg.py:109,128, without a BatchNorm behind them).  One workgroup per 512 rows,
// float4 columns x row lanes, partial sums combined in LDS, one partial row per workgroup into `part` [nblocks][C]; a second
// small launch adds the partial rows in a FIXED order (no atomics: the same bits from run to run, as torch's reduction gave;
// torch's reduction took 17 - 23 us for 12 - 50 MB: nine launches per DGCNN step).
constexpr int CS_ROWS = 512;
__global__ __launch_bounds__(256) void col_sum_kernel(const float *__restrict__ Y, long long ld, int P, int C, float *__restrict__ part)
{
    __shared__ float4 s_red[256];
    const int C4 = C >> 2;
    const int r_begin = blockIdx.x * CS_ROWS, r_end = min(P, r_begin + CS_ROWS);
    for (int cbase = 0; cbase < C4; cbase += 256) {
        const int cols = min(256, C4 - cbase);
        const int lanes = 256 / cols;
        const int c4 = cbase + threadIdx.x % cols, rl = threadIdx.x / cols;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;     // two partials per thread: independent adds
        if (rl < lanes) {
            int r = r_begin + rl;
            for (; r + lanes < r_end; r += 2 * lanes) {
                const float4 u = ld4g(Y + (size_t)r * ld + 4 * c4), v = ld4g(Y + (size_t)(r + lanes) * ld + 4 * c4);
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
            }
            if (r < r_end) {
                const float4 u = ld4g(Y + (size_t)r * ld + 4 * c4);
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            }
        }
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        s_red[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < cols) {
            float4 t = s_red[threadIdx.x];
            for (int l = 1; l < lanes; ++l) {
                const float4 u = s_red[threadIdx.x + l * cols];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4 *>(part + (size_t)blockIdx.x * C + 4 * c4) = t;
        }
        __syncthreads();
    }
}

// out[c] = sum over the partial rows, four interleaved chains per column in a fixed order (blockIdx.y: one output row per nb
// consecutive partial rows -- the per-sample sums of prifit_col_sum_samples)
__global__ __launch_bounds__(256) void col_sum_reduce_kernel(const float *__restrict__ part, int nb, int C, float *__restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    part += (size_t)blockIdx.y * nb * C;
    out += (size_t)blockIdx.y * C;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 3 < nb; b += 4) {
        a0 += part[(size_t)b * C + c]; a1 += part[(size_t)(b + 1) * C + c];
        a2 += part[(size_t)(b + 2) * C + c]; a3 += part[(size_t)(b + 3) * C + c];
    }
    for (; b < nb; ++b) a0 += part[(size_t)b * C + c];
    out[c] = (a0 + a1) + (a2 + a3);
}

// Column sum / sum of squares of a matrix (used when the producer was not a GEMM with fused stats).
__global__ __launch_bounds__(256) void col_stats_kernel(const float *__restrict__ Y, long long ld, int P, int C,
                                                        float *__restrict__ slab)
{
    column_reduce(P, C, slab, [&](int r, int c4, float4 &a0, float4 &a1) {
        const float4 y = ld4g(Y + (size_t)r * ld + 4 * c4);
        a0.x += y.x; a0.y += y.y; a0.z += y.z; a0.w += y.w;
        a1.x += y.x * y.x; a1.y += y.y * y.y; a1.z += y.z * y.z; a1.w += y.w * y.w;
    });
}

// Activation with a negative slope (0: ReLU, 0.2: the LeakyReLU of src/dgcnn.py:162).
__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }
// Row offset into the per-sample coefficient tables [samples][C]: rps rows share one table row
// (rps == 0: one table for everything = BatchNorm; rps = rows per sample = GroupNorm).
__device__ __forceinline__ long long tab_off(long long row, int rps, int C) { return rps ? (row / rps) * C : 0; }

// out = act(Y * scale + shift)
__global__ __launch_bounds__(256) void affine_relu_kernel(const float *__restrict__ Y, long long ldy,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int P, int C4, int rps,
                                                          float slope, float *__restrict__ out, long long ldo)
{
    const long long total = (long long)P * C4;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long r = id / C4;
        const int c = (int)(id - r * C4) * 4;
        const long long to = tab_off(r, rps, C4 * 4) + c;
        const float4 y = ld4g(Y + r * ldy + c), s = ld4g(scale + to), t = ld4g(shift + to);
        st4g(out + r * ldo + c, make_float4(act(fmaf(y.x, s.x, t.x), slope), act(fmaf(y.y, s.y, t.y), slope),
                                            act(fmaf(y.z, s.z, t.z), slope), act(fmaf(y.w, s.w, t.w), slope)));
    }
}

// Group max-pool of relu(bn(Y)) over the K samples of each group (torch.max(new_points, 2)[0]).
// Y [G*K, ld]; out [G, ldo] (+col0); arg [G, C] = winning k (first maximum).
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float *__restrict__ Y, long long ldy,
                                                       const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int G, int K, int C4,
                                                       int rps, float slope, float *__restrict__ out,
                                                       long long ldo, int32_t *__restrict__ arg)
{
    const long long total = (long long)G * C4;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long gidx = id / C4;
        const int c = (int)(id - gidx * C4) * 4;
        const long long to = tab_off(gidx * K, rps, C4 * 4) + c;
        const float4 s = ld4g(scale + to), t = ld4g(shift + to);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 bi = make_int4(0, 0, 0, 0);
        const float *row = Y + gidx * K * ldy + c;
        for (int k0 = 0; k0 < K; k0 += 8) {  // eight independent row loads in flight, then the ordered compares
            float4 yv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) yv[j] = ld4g(row + (long long)(k0 + j < K ? k0 + j : K - 1) * ldy);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                if (k >= K) break;
                const float4 y = yv[j];
                const float vx = fmaf(y.x, s.x, t.x), vy = fmaf(y.y, s.y, t.y), vz = fmaf(y.z, s.z, t.z),
                            vw = fmaf(y.w, s.w, t.w);
                if (vx > best.x) { best.x = vx; bi.x = k; }
                if (vy > best.y) { best.y = vy; bi.y = k; }
                if (vz > best.z) { best.z = vz; bi.z = k; }
                if (vw > best.w) { best.w = vw; bi.w = k; }
            }
        }
        st4g(out + gidx * ldo + c,
             make_float4(act(best.x, slope), act(best.y, slope), act(best.z, slope), act(best.w, slope)));
        *reinterpret_cast<int4 *>(arg + gidx * (C4 * 4) + c) = bi;
    }
}

// ---------------------------------------------------------------------------------------------
// first layer of a set-abstraction MLP by linearity:  W1 [feat_j | xyz_j - c_g] = U_j - Vc_g with
// U = [feat | xyz] W1^T per POINT and Vc = c W1x^T per CENTRE (two small GEMMs), so the per-sample work is
// a gather of C1-wide rows instead of a (C+3)-wide gather plus a GEMM over all grouped samples.
//   Y1[(b,s,k), :] = U[b, idx[b,s,k], :] - Vc[b,s,:] + bias        (+ per-block column sum / sum-of-squares
//   slabs for the BatchNorm that follows, same layout as the GEMM epilogue's with 512-row slabs)
// ---------------------------------------------------------------------------------------------
constexpr int GL_UNR = 4;
__global__ __launch_bounds__(256) void gather_linear_fwd_kernel(const float *__restrict__ U,
                                                                const float *__restrict__ Vc,
                                                                const float *__restrict__ bias,
                                                                const int32_t *__restrict__ idx, int N, int S, int K,
                                                                int P, int C, float *__restrict__ Y,
                                                                float *__restrict__ slab)
{
    __shared__ float4 s_red[2][256];
    __shared__ int s_src[RED_ROWS];   // row of U (b*N + n), or -1 for an out-of-range index
    __shared__ int s_grp[RED_ROWS];   // row of Vc (b*S + s)
    const int C4 = C >> 2;
    const int r_begin = blockIdx.x * RED_ROWS, r_end = min(P, r_begin + RED_ROWS);
    for (int i = threadIdx.x; i < r_end - r_begin; i += 256) {
        const int r = r_begin + i, g = r / K, n = idx[r];
        s_grp[i] = g;
        s_src[i] = (n >= 0 && n < N) ? (g / S) * N + n : -1;
    }
    __syncthreads();
    for (int cbase = 0; cbase < C4; cbase += 256) {
        const int cols = min(256, C4 - cbase);
        const int lanes = 256 / cols;
        const int c4 = cbase + threadIdx.x % cols, rl = threadIdx.x / cols;
        const int c = 4 * c4;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        const float4 bb = bias ? ld4g(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        auto emit = [&](int i, float4 u, float4 v, bool ok) {
            float4 y = ok ? make_float4(u.x - v.x, u.y - v.y, u.z - v.z, u.w - v.w) : make_float4(0.f, 0.f, 0.f, 0.f);
            y.x += bb.x; y.y += bb.y; y.z += bb.z; y.w += bb.w;
            st4g(Y + (size_t)(r_begin + i) * C + c, y);
            a0.x += y.x; a0.y += y.y; a0.z += y.z; a0.w += y.w;
            a1.x += y.x * y.x; a1.y += y.y * y.y; a1.z += y.z * y.z; a1.w += y.w * y.w;
        };
        if (rl < lanes) {
            const int nrow = r_end - r_begin;
            int i = rl;
            for (; i + (GL_UNR - 1) * lanes < nrow; i += GL_UNR * lanes) {  // GL_UNR independent row loads in flight
                float4 u[GL_UNR], v[GL_UNR];
                bool ok[GL_UNR];
#pragma unroll
                for (int j = 0; j < GL_UNR; ++j) {
                    const int src = s_src[i + j * lanes];
                    ok[j] = src >= 0;
                    u[j] = ld4g(U + (size_t)(ok[j] ? src : 0) * C + c);
                    v[j] = ld4g(Vc + (size_t)s_grp[i + j * lanes] * C + c);
                }
#pragma unroll
                for (int j = 0; j < GL_UNR; ++j) emit(i + j * lanes, u[j], v[j], ok[j]);
            }
            for (; i < nrow; i += lanes) {
                const int src = s_src[i];
                emit(i, ld4g(U + (size_t)(src >= 0 ? src : 0) * C + c), ld4g(Vc + (size_t)s_grp[i] * C + c), src >= 0);
            }
        }
        s_red[0][threadIdx.x] = a0;
        s_red[1][threadIdx.x] = a1;
        __syncthreads();
        if (threadIdx.x < cols) {
            float4 t0 = s_red[0][threadIdx.x], t1 = s_red[1][threadIdx.x];
            for (int l = 1; l < lanes; ++l) {
                const float4 u0 = s_red[0][threadIdx.x + l * cols], u1 = s_red[1][threadIdx.x + l * cols];
                t0.x += u0.x; t0.y += u0.y; t0.z += u0.z; t0.w += u0.w;
                t1.x += u1.x; t1.y += u1.y; t1.z += u1.z; t1.w += u1.w;
            }
            st4g(slab + ((size_t)blockIdx.x * 2 + 0) * C + c, t0);
            st4g(slab + ((size_t)blockIdx.x * 2 + 1) * C + c, t1);
        }
        __syncthreads();
    }
}

// autograd: dU[b, idx, :] += dY (float atomics, one lane per float: 256 B contiguous per wave-instruction);
// dVc[g, :] = -sum_k dY[(g,k), :]
__global__ __launch_bounds__(256) void gather_linear_bwd_kernel(const float *__restrict__ dY,
                                                                const int32_t *__restrict__ idx, int N, int S, int K,
                                                                int C, long long total, float *__restrict__ dU,
                                                                float *__restrict__ dVc)
{
    constexpr int UNR = 8;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long g = id / C;   // b*S + s
        const int c = (int)(id - g * C);
        const int b = (int)(g / S);
        const float *src = dY + (size_t)g * K * C + c;
        const int32_t *ix = idx + (size_t)g * K;
        float *dst = dU + (size_t)b * N * C + c;
        float acc = 0.f;
        int k = 0;
        for (; k + UNR <= K; k += UNR) {
            float v[UNR];
            int n[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) { v[j] = src[(size_t)(k + j) * C]; n[j] = ix[k + j]; }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                if (n[j] >= 0 && n[j] < N) unsafeAtomicAdd(dst + (size_t)n[j] * C, v[j]);
                acc += v[j];
            }
        }
        for (; k < K; ++k) {
            const float v = src[(size_t)k * C];
            const int n = ix[k];
            if (n >= 0 && n < N) unsafeAtomicAdd(dst + (size_t)n * C, v);
            acc += v;
        }
        dVc[id] = -acc;
    }
}

// The same autograd with the BatchNorm + ReLU backward of the gathered layer folded in and the scatter staged in LDS:
//     dY = a (Y s + t > 0 ? G : 0) + (b Y + d)         (bn_relu_bwd_apply_kernel's expression, term by term)
//     dU[b, idx, :] += dY,   dVc[g, :] -= sum_k dY[(g,k), :]
// One workgroup = (shape, a range of GLB_PTS points, a range of centres): it keeps dU of its points in LDS ([GLB_PTS][C]
// floats), walks the index lists of its centres, loads the rows whose index falls into its range -- whole rows: one wave
// instruction = the 4 C contiguous bytes of a row, lane = 2 channels -- forms dY and adds it there (each wave owns an eighth
// of the points: plain read-modify-write); at the end the non-zero entries go to global memory with float atomics.  Every row is loaded by exactly one workgroup, the global
// atomics drop from one per element of dY (B S K C: 75 M at SA2) to at most one per element of dU and centre range, and
// the pass that wrote dY (read G, read Y, write dY) is gone.  (Splitting the CHANNELS over workgroups instead -- 128-byte
// pieces of every row read by four different workgroups -- measured slower than the pass it replaces.)
// (Round 5, measured and NOT kept: eight waves per workgroup instead of four -- twice the rows in flight per CU -- 96 -> 117 us
// at K = 64, 163 -> 154 us at K = 128, stand-alone, alternating builds: -DGLB_WAVES_N=8.)
#ifndef GLB_WAVES_N
#define GLB_WAVES_N 4
#endif
#ifndef GLB_PROBE
#define GLB_PROBE 0     // timing-only diagnosis builds (wrong results): 1 no global atomics, 2 no row loads, 3 no LDS update
#endif
constexpr int GLB_PTS = 128, GLB_CMAX = 128, GLB_WAVES = GLB_WAVES_N, GLB_NTH = 64 * GLB_WAVES;
__global__ __launch_bounds__(GLB_NTH, 2) void gather_linear_bwd_bn_kernel(
    const float *__restrict__ G, const float *__restrict__ Y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ ca, const float *__restrict__ cb,
    const float *__restrict__ cd, const int32_t *__restrict__ idx, int N, int S, int K, int C, int splits,
    float *__restrict__ dU, float *__restrict__ dVc)
{
    __shared__ float s_acc[GLB_PTS * GLB_CMAX];
    __shared__ int s_rows[GLB_WAVES][128];  // per wave: the rows of the current batch of the index list that are ours
    const int n0 = blockIdx.x * GLB_PTS, split = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = 2 * lane;                 // this lane's channel pair
    const bool cok = c < C;
    for (int i = threadIdx.x; i < GLB_PTS * C; i += GLB_NTH) s_acc[i] = 0.f;
    float2 sc = {0.f, 0.f}, sh = sc, a = sc, bb = sc, d = sc;
    if (cok) {
        sc = *reinterpret_cast<const float2 *>(scale + c); sh = *reinterpret_cast<const float2 *>(shift + c);
        a = *reinterpret_cast<const float2 *>(ca + c); bb = *reinterpret_cast<const float2 *>(cb + c);
        d = *reinterpret_cast<const float2 *>(cd + c);
    }
    __syncthreads();
    const int per = (S + splits - 1) / splits;
    const int s_begin = split * per, s_end = min(S, s_begin + per);
    int *rows = s_rows[wave];
    // every wave walks ALL index lists of the workgroup's centres and takes the rows of ITS points (point % GLB_WAVES == wave): a point's
    // accumulator is then touched by one wave only and the LDS update needs no atomic (LDS float atomics measured ~2 cycles
    // per LANE: 75 M of them were 3 x the time of the loads).
    // Round 5: the walk is over CHUNKS of 64 slots (centre, k0) with the index chunk of step t + 1 requested before the rows of
    // step t are processed -- a centre used to cost two dependent memory round trips (its indices, then the ~4..8 rows that
    // turn out to be this wave's), and with ~26 centres per workgroup that chain WAS the kernel's time.
    const int nblk = (K + 63) >> 6;
    const int T = (s_end - s_begin) * nblk;
    auto load_chunk = [&](int t) -> int {
        const int sidx = s_begin + t / nblk, r = ((t % nblk) << 6) + lane;
        return (t < T && r < K) ? idx[((size_t)b * S + sidx) * K + r] : -1;
    };
    int n_cur = T > 0 ? load_chunk(0) : -1;
    float2 acc = {0.f, 0.f}, pacc = {0.f, 0.f};
    int first = 0, pad_seen = 0;
    bool pads_here = false;
    const float *gp = G, *yp = Y;
    long long g = 0;
    for (int t = 0; t < T; ++t) {
        const int n_nxt = load_chunk(t + 1);
        __builtin_amdgcn_sched_barrier(0);      // (the request stays HERE: left alone the scheduler sinks it to its use)
        const int k0 = (t % nblk) << 6;
        if (k0 == 0) {
            g = (long long)b * S + s_begin + t / nblk;
            gp = G + (size_t)g * K * C + c; yp = Y + (size_t)g * K * C + c;
            acc = make_float2(0.f, 0.f); pacc = make_float2(0.f, 0.f);
            // Padding: a ball with fewer than K points repeats its FIRST index (models/pointnet_util.py:104-106), on sparse clouds
            // in most of the slots -- all of them rows of one point, i.e. of one wave.  The workgroup that owns that point deals
            // the padded slots round-robin over its waves instead; their sum stays in registers and goes to dU with one
            // global atomic per channel and group.
            first = __builtin_amdgcn_readfirstlane(n_cur);
            pads_here = first >= n0 && first < n0 + GLB_PTS && first < N;
            pad_seen = 0;
        }
        {
            // the rows of this chunk that are this wave's: compacted in slot order (ballot + prefix count)
            const int r = k0 + lane;
            const int n = n_cur;
            const bool pad = pads_here && r > 0 && r < K && n == first;
            const unsigned long long pm = __ballot(pad);
            const bool my_pad = pad && ((pad_seen + __popcll(pm & ((1ull << lane) - 1ull))) & (GLB_WAVES - 1)) == wave;
            pad_seen += __popcll(pm);
            const bool mine = my_pad || (!pad && n >= n0 && n < n0 + GLB_PTS && n < N && (n & (GLB_WAVES - 1)) == wave);
            const unsigned long long m = __ballot(mine);
            if (mine) rows[__popcll(m & ((1ull << lane) - 1ull))] = (r << 8) | (my_pad ? 0x80 : 0) | ((n - n0) & 0x7f);
            const int cnt = __popcll(m);
            // (wave-private LDS: the writes above are visible to the wave's own later reads without a barrier)
            __builtin_amdgcn_wave_barrier();
            constexpr int UNR = 8;
            for (int q = 0; q < cnt; q += UNR) {
                float2 gv[UNR], yv[UNR];
                int pk[UNR];
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    pk[j] = rows[min(q + j, cnt - 1)];
                    const size_t off = (size_t)(pk[j] >> 8) * C;
                    gv[j] = (cok && GLB_PROBE != 2) ? *reinterpret_cast<const float2 *>(gp + off) : make_float2(0.f, 0.f);
                    yv[j] = (cok && GLB_PROBE != 2) ? *reinterpret_cast<const float2 *>(yp + off) : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < UNR; ++j) {
                    if (q + j < cnt && cok) {
                        const float dx = fmaf(a.x, fmaf(yv[j].x, sc.x, sh.x) > 0.f ? gv[j].x : 0.f, fmaf(bb.x, yv[j].x, d.x));
                        const float dyy = fmaf(a.y, fmaf(yv[j].y, sc.y, sh.y) > 0.f ? gv[j].y : 0.f, fmaf(bb.y, yv[j].y, d.y));
                        if ((pk[j] & 0x80) || GLB_PROBE == 3) {   // (wave-uniform) a padded slot: summed in registers
                            pacc.x += dx; pacc.y += dyy;
                        } else {
                            // plain read-modify-write: the point is this wave's alone, and a wave's LDS accesses execute in order
                            float2 *dst = reinterpret_cast<float2 *>(s_acc + (pk[j] & 0x7f) * C + c);
                            float2 v = *dst;
                            v.x += dx; v.y += dyy;
                            *dst = v;
                        }
                        acc.x += dx; acc.y += dyy;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (k0 + 64 >= K && GLB_PROBE != 1) {      // the centre's last chunk: its sums leave
            if (cok && (acc.x != 0.f || acc.y != 0.f)) {
                unsafeAtomicAdd(dVc + (size_t)g * C + c, -acc.x);
                unsafeAtomicAdd(dVc + (size_t)g * C + c + 1, -acc.y);
            }
            if (cok && (pacc.x != 0.f || pacc.y != 0.f)) {
                unsafeAtomicAdd(dU + ((size_t)b * N + first) * C + c, pacc.x);
                unsafeAtomicAdd(dU + ((size_t)b * N + first) * C + c + 1, pacc.y);
            }
        }
        n_cur = n_nxt;
    }
    __syncthreads();
    const int npts = min(GLB_PTS, N - n0);
    float *dst = dU + ((size_t)b * N + n0) * C;
    for (int i = threadIdx.x; i < npts * C; i += GLB_NTH) {
        const float v = s_acc[i];
        if (v != 0.f && GLB_PROBE != 1) unsafeAtomicAdd(dst + i, v);
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// Partials of m1 = sum(G * mask), m2 = sum(G * mask * yhat) with mask = (scale*Y+shift > 0),
// yhat = (Y - mean) * invstd.
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const float *__restrict__ Gr, long long ldg,
                                                                 const float *__restrict__ Y, long long ldy,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd, int P, int C,
                                                                 int rps, float slope, float *__restrict__ slab, const BnTail tail)
{
    column_reduce(P, C, slab, [&](int r, int c4, float4 &a0, float4 &a1) {
        const int c = 4 * c4;
        const long long to = tab_off(r, rps, C) + c;
        const float4 g = ld4g(Gr + (size_t)r * ldg + c), y = ld4g(Y + (size_t)r * ldy + c);
        const float4 s = ld4g(scale + to), t = ld4g(shift + to), mu = ld4g(mean + to), is = ld4g(invstd + to);
        const float gx = fmaf(y.x, s.x, t.x) > 0.f ? g.x : g.x * slope, gy = fmaf(y.y, s.y, t.y) > 0.f ? g.y : g.y * slope,
                    gz = fmaf(y.z, s.z, t.z) > 0.f ? g.z : g.z * slope, gw = fmaf(y.w, s.w, t.w) > 0.f ? g.w : g.w * slope;
        a0.x += gx; a0.y += gy; a0.z += gz; a0.w += gw;
        a1.x += gx * ((y.x - mu.x) * is.x); a1.y += gy * ((y.y - mu.y) * is.y);
        a1.z += gz * ((y.z - mu.z) * is.z); a1.w += gw * ((y.w - mu.w) * is.w);
    }, tail);
}

// Same partials when the gradient arrives through the group max-pool: only the winning sample of
// each (group, channel) carries gradient gp[g, c] (and only if the pooled value is > 0).
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(const float *__restrict__ gp, long long ldgp,
                                                              const float *__restrict__ Y, long long ldy,
                                                              const int32_t *__restrict__ arg,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, int G, int K,
                                                              int C, int rps, float slope,
                                                              float *__restrict__ slab, const BnTail tail)
{
    column_reduce<POOL_RED_ROWS>(G, C, slab, [&](int gi, int c4, float4 &a0, float4 &a1) {
        const int c = 4 * c4;
        const long long to = tab_off((long long)gi * K, rps, C) + c;
        const float4 g = ld4g(gp + (size_t)gi * ldgp + c);
        const int4 k = *reinterpret_cast<const int4 *>(arg + (size_t)gi * C + c);
        const float *base = Y + (size_t)gi * K * ldy + c;
        const float yx = base[(size_t)k.x * ldy], yy = base[(size_t)k.y * ldy + 1],
                    yz = base[(size_t)k.z * ldy + 2], yw = base[(size_t)k.w * ldy + 3];
        const float4 s = ld4g(scale + to), t = ld4g(shift + to), mu = ld4g(mean + to), is = ld4g(invstd + to);
        const float gx = fmaf(yx, s.x, t.x) > 0.f ? g.x : g.x * slope, gy = fmaf(yy, s.y, t.y) > 0.f ? g.y : g.y * slope,
                    gz = fmaf(yz, s.z, t.z) > 0.f ? g.z : g.z * slope, gw = fmaf(yw, s.w, t.w) > 0.f ? g.w : g.w * slope;
        a0.x += gx; a0.y += gy; a0.z += gz; a0.w += gw;
        a1.x += gx * ((yx - mu.x) * is.x); a1.y += gy * ((yy - mu.y) * is.y);
        a1.z += gz * ((yz - mu.z) * is.z); a1.w += gw * ((yw - mu.w) * is.w);
    }, tail);
}

// m1, m2 -> dgamma = m2, dbeta = m1 and the coefficients of dY = a * Gmasked + b * Y + d:
//   train: dY = gamma*invstd * (Gm - m1/n - yhat * m2/n)
//   eval : dY = gamma*invstd_running * Gm            (b = d = 0)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ slab, int nslab, int C,
                                                              double count, int training,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd,
                                                              float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta, float *__restrict__ ca,
                                                              float *__restrict__ cb, float *__restrict__ cd)
{
    __shared__ double s_s[4], s_q[4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nslab; i += 256) {
        s += (double)slab[((size_t)i * 2 + 0) * C + c];
        q += (double)slab[((size_t)i * 2 + 1) * C + c];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if ((threadIdx.x & 63) == 0) { s_s[threadIdx.x >> 6] = s; s_q[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m1 = s_s[0] + s_s[1] + s_s[2] + s_s[3];
        const double m2 = s_q[0] + s_q[1] + s_q[2] + s_q[3];
        float o[5];
        bn_bwd_coefs(m1, m2, count, training, scale[c], mean[c], invstd[c], o, 1);
        dgamma[c] = o[0];
        dbeta[c] = o[1];
        ca[c] = o[2];
        cb[c] = o[3];
        cd[c] = o[4];
    }
}

// dY = a * (G masked by the ReLU) + b * Y + d
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float *__restrict__ Gr, long long ldg,
                                                                const float *__restrict__ Y, long long ldy,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift,
                                                                const float *__restrict__ ca,
                                                                const float *__restrict__ cb,
                                                                const float *__restrict__ cd, int P, int C4,
                                                                int rps, float slope, float *__restrict__ dY,
                                                                long long ldd)
{
    const long long total = (long long)P * C4;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long r = id / C4;
        const int c = (int)(id - r * C4) * 4;
        const long long to = tab_off(r, rps, C4 * 4) + c;
        const float4 g = ld4g(Gr + r * ldg + c), y = ld4g(Y + r * ldy + c);
        const float4 s = ld4g(scale + to), t = ld4g(shift + to), a = ld4g(ca + to), b = ld4g(cb + to), d = ld4g(cd + to);
        float4 o;
        o.x = fmaf(a.x, fmaf(y.x, s.x, t.x) > 0.f ? g.x : g.x * slope, fmaf(b.x, y.x, d.x));
        o.y = fmaf(a.y, fmaf(y.y, s.y, t.y) > 0.f ? g.y : g.y * slope, fmaf(b.y, y.y, d.y));
        o.z = fmaf(a.z, fmaf(y.z, s.z, t.z) > 0.f ? g.z : g.z * slope, fmaf(b.z, y.z, d.z));
        o.w = fmaf(a.w, fmaf(y.w, s.w, t.w) > 0.f ? g.w : g.w * slope, fmaf(b.w, y.w, d.w));
        st4g(dY + r * ldd + c, o);
    }
}

// dY[g,k,c] = a * (k == arg[g,c] && pooled > 0 ? gp[g,c] : 0) + b * Y + d
__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(const float *__restrict__ gp, long long ldgp,
                                                             const float *__restrict__ Y, long long ldy,
                                                             const int32_t *__restrict__ arg,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift,
                                                             const float *__restrict__ ca,
                                                             const float *__restrict__ cb,
                                                             const float *__restrict__ cd, int G, int K, int C4,
                                                             int rps, float slope, float *__restrict__ dY,
                                                             long long ldd)
{
    const long long total = (long long)G * K * C4;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long r = id / C4;  // g*K + k
        const int c = (int)(id - r * C4) * 4;
        const long long gi = r / K;
        const int k = (int)(r - gi * K);
        const long long to = tab_off(r, rps, C4 * 4) + c;
        const float4 g = ld4g(gp + gi * ldgp + c), y = ld4g(Y + r * ldy + c);
        const int4 w = *reinterpret_cast<const int4 *>(arg + gi * (C4 * 4) + c);
        const float4 s = ld4g(scale + to), t = ld4g(shift + to), a = ld4g(ca + to), b = ld4g(cb + to), d = ld4g(cd + to);
        float4 o;
        o.x = fmaf(a.x, k == w.x ? (fmaf(y.x, s.x, t.x) > 0.f ? g.x : g.x * slope) : 0.f, fmaf(b.x, y.x, d.x));
        o.y = fmaf(a.y, k == w.y ? (fmaf(y.y, s.y, t.y) > 0.f ? g.y : g.y * slope) : 0.f, fmaf(b.y, y.y, d.y));
        o.z = fmaf(a.z, k == w.z ? (fmaf(y.z, s.z, t.z) > 0.f ? g.z : g.z * slope) : 0.f, fmaf(b.z, y.z, d.z));
        o.w = fmaf(a.w, k == w.w ? (fmaf(y.w, s.w, t.w) > 0.f ? g.w : g.w * slope) : 0.f, fmaf(b.w, y.w, d.w));
        st4g(dY + r * ldd + c, o);
    }
}

// The same for C / 4 dividing 256: a thread keeps ITS four channels -- the five coefficient vectors are loaded once (BatchNorm)
// or once per sample (GroupNorm tables), not per element -- and walks rows with 32-bit arithmetic (the generic form pays two
// 64-bit divisions and five 16-byte table loads per 16 bytes of Y: 4.1 TB/s where the BatchNorm apply pass reaches 5.5).
__global__ __launch_bounds__(256) void pool_bwd_apply_rows_kernel(const float *__restrict__ gp, long long ldgp,
                                                                  const float *__restrict__ Y, long long ldy,
                                                                  const int32_t *__restrict__ arg,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ shift,
                                                                  const float *__restrict__ ca,
                                                                  const float *__restrict__ cb,
                                                                  const float *__restrict__ cd, int G, int K, int C4,
                                                                  int rps, float slope, float *__restrict__ dY, long long ldd)
{
    // a thread walks the K rows of a pooling GROUP on its four channels: the pooled gradient and the winner's index are
    // loaded once per group, the row loop is loads, seven vector operations per float and stores -- four rows in flight
    const int gpb = 256 / C4;                         // groups per block and trip
    const int c = (threadIdx.x % C4) * 4;
    float4 s = ld4g(scale + c), t = ld4g(shift + c), a = ld4g(ca + c), b = ld4g(cb + c), d = ld4g(cd + c);
    int cur = 0;                                      // rps != 0 (GroupNorm): one table row per sample of rps rows
    for (int gi = blockIdx.x * gpb + threadIdx.x / C4; gi < G; gi += gridDim.x * gpb) {
        if (rps) {
            const int smp = (int)(((long long)gi * K) / rps);
            if (smp != cur) {
                cur = smp;
                const long long to = (long long)smp * (C4 * 4) + c;
                s = ld4g(scale + to); t = ld4g(shift + to); a = ld4g(ca + to); b = ld4g(cb + to); d = ld4g(cd + to);
            }
        }
        const float4 g = ld4g(gp + (long long)gi * ldgp + c);
        const int4 w = *reinterpret_cast<const int4 *>(arg + (long long)gi * (C4 * 4) + c);
        const float *yp = Y + (long long)gi * K * ldy + c;
        float *dp = dY + (long long)gi * K * ldd + c;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float4 y = ld4g(yp + (long long)k * ldy);
            float4 o;
            o.x = fmaf(a.x, k == w.x ? (fmaf(y.x, s.x, t.x) > 0.f ? g.x : g.x * slope) : 0.f, fmaf(b.x, y.x, d.x));
            o.y = fmaf(a.y, k == w.y ? (fmaf(y.y, s.y, t.y) > 0.f ? g.y : g.y * slope) : 0.f, fmaf(b.y, y.y, d.y));
            o.z = fmaf(a.z, k == w.z ? (fmaf(y.z, s.z, t.z) > 0.f ? g.z : g.z * slope) : 0.f, fmaf(b.z, y.z, d.z));
            o.w = fmaf(a.w, k == w.w ? (fmaf(y.w, s.w, t.w) > 0.f ? g.w : g.w * slope) : 0.f, fmaf(b.w, y.w, d.w));
            st4g(dp + (long long)k * ldd, o);
        }
    }
}

// Backward of a max-pooled, normalised by-linearity layer in ONE pass (the DGCNN edge convolution, src/dgcnn.py:98-107 +
// :157-171: y[(g,k)] = U[idx[g,k]] - Vc[g], GroupNorm / BatchNorm, LeakyReLU, max over the K rows of group g):
//   dy = a * (k == arg[g,c] ? act'(y) * gp[g,c] : 0) + b * y + d          (pool_bwd_apply's expression, term by term)
//   dU[idx[g,k]] += dy (one float atomic per element: a wave instruction covers 256 contiguous bytes, the full-rate form),
//   dVc[g] = -sum_k dy.
// The dY tensor ([B N k, C]: 0.25 - 0.5 GB per layer) is neither written nor read back.  One thread = (group, channel).
__global__ __launch_bounds__(256) void gather_linear_bwd_pool_kernel(
    const float *__restrict__ gp, long long ldgp, const float *__restrict__ Y, const int32_t *__restrict__ arg,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ ca,
    const float *__restrict__ cb, const float *__restrict__ cd, const int32_t *__restrict__ idx, int N, int S, int K, int C,
    int rps, float slope, long long total, float *__restrict__ dU, float *__restrict__ dVc)
{
    constexpr int UNR = 4;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long g = id / C;   // b*S + s
        const int c = (int)(id - g * C);
        const int b = (int)(g / S);
        const long long to = (rps ? ((g * K) / rps) * C : 0) + c;       // per-sample table row (GroupNorm) or the one row
        const float s = scale[to], t = shift[to], a = ca[to], bb = cb[to], d = cd[to];
        const float gv = gp[g * ldgp + c];
        const int w = arg[g * C + c];
        const float *src = Y + (size_t)g * K * C + c;
        const int32_t *ix = idx + (size_t)g * K;
        float *dst = dU + (size_t)b * N * C + c;
        float acc = 0.f;
        int k = 0;
        for (; k + UNR <= K; k += UNR) {
            float y[UNR];
            int n[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) { y[j] = src[(size_t)(k + j) * C]; n[j] = ix[k + j]; }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const float v = fmaf(a, (k + j) == w ? (fmaf(y[j], s, t) > 0.f ? gv : gv * slope) : 0.f, fmaf(bb, y[j], d));
                if (n[j] >= 0 && n[j] < N) unsafeAtomicAdd(dst + (size_t)n[j] * C, v);
                acc += v;
            }
        }
        for (; k < K; ++k) {
            const float y = src[(size_t)k * C];
            const int n = ix[k];
            const float v = fmaf(a, k == w ? (fmaf(y, s, t) > 0.f ? gv : gv * slope) : 0.f, fmaf(bb, y, d));
            if (n >= 0 && n < N) unsafeAtomicAdd(dst + (size_t)n * C, v);
            acc += v;
        }
        dVc[id] = -acc;
    }
}

// T[g,c] = a[c] * (relu'(y at the winning sample) * gp[g,c]): the sparse term of the pooled layer's dY
// (dY[g,k,c] = T[g,c] * [k == arg[g,c]] + b[c] * Y + d[c]) for consumers that form dY in their operand staging.
__global__ __launch_bounds__(256) void pool_bwd_table_kernel(const float *__restrict__ gp, long long ldgp,
                                                             const float *__restrict__ Y, long long ldy,
                                                             const int32_t *__restrict__ arg,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift,
                                                             const float *__restrict__ ca, int G, int K, int C,
                                                             float slope, float *__restrict__ T)
{
    const long long total = (long long)G * C;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long g = id / C;
        const int c = (int)(id - g * C);
        const int w = arg[id];
        const float y = Y[(g * K + w) * ldy + c];
        const float gv = gp[g * ldgp + c];
        T[id] = ca[c] * (fmaf(y, scale[c], shift[c]) > 0.f ? gv : gv * slope);
    }
}

// Group max-pool from the per-32-row candidates of prifit_gemm_stream_pool_f32 (max, argmax, min, argmin of the raw
// pre-activations): max_k relu(s*y+t) = relu(s*y*+t) with y* the block maximum for s > 0, the block minimum for s < 0.
// Ties go to the lowest sample, like torch.max / pool_fwd_kernel (which compares s*y+t; the two can only disagree on
// WHICH of two samples with equal activation is reported, or when the pooled activation is <= 0 and carries no gradient).
__global__ __launch_bounds__(256) void pool_from_candidates_kernel(const float *__restrict__ cand,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift, int G, int K, int C,
                                                                   int rps, float slope, float *__restrict__ out, long long ldo,
                                                                   int32_t *__restrict__ arg, float *__restrict__ ystar)
{
    const long long total = (long long)G * C;
    const int nb = K >> 5;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long gidx = id / C;
        const int c = (int)(id - gidx * C);
        const long long to = tab_off(gidx * K, rps, C) + c;   // rps != 0: per-sample coefficient rows (GroupNorm)
        const float s = scale[to], t = shift[to];
        const float *cd = cand + gidx * nb * 4 * C + c;
        float best;
        int bi = 0;
        if (s > 0.f) {
            best = -INFINITY;
            for (int b = 0; b < nb; ++b) {
                const float v = cd[(long long)(b * 4 + 0) * C];
                if (v > best) { best = v; bi = b * 32 + __float_as_int(cd[(long long)(b * 4 + 1) * C]); }
            }
        } else if (s < 0.f) {
            best = INFINITY;
            for (int b = 0; b < nb; ++b) {
                const float v = cd[(long long)(b * 4 + 2) * C];
                if (v < best) { best = v; bi = b * 32 + __float_as_int(cd[(long long)(b * 4 + 3) * C]); }
            }
        } else {
            best = cd[0];  // every sample has the activation t: the first one wins
        }
        out[gidx * ldo + c] = act(fmaf(best, s, t), slope);
        arg[id] = bi;
        if (ystar) ystar[id] = best;     // the winner's pre-activation: what the backward needs of Y when Y is not stored
    }
}

static inline int ew_grid(long long total)
{
    long long g = (total + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > 256 * 32 ? 256 * 32 : g);
}
static inline bool bad_mat(const void *p, long long ld, int C)
{
    return !p || (C & 3) || (ld & 3) || ((uintptr_t)p & 15) || ld < C;
}

extern "C" {

int prifit_reduce_rows_per_slab(void) { return RED_ROWS; }
int prifit_pool_reduce_groups_per_slab(void) { return POOL_RED_ROWS; }

int prifit_col_stats(const float *Y, long long ld, int P, int C, float *slab, void *stream)
{
    if (bad_mat(Y, ld, C) || !slab || P <= 0 || C <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(col_stats_kernel, dim3((P + RED_ROWS - 1) / RED_ROWS), dim3(256), 0, as_stream(stream), Y,
                       ld, P, C, slab);
    return prifit_check_launch();
}

int prifit_bn_finalize(const float *slab, int nslab, int C, double count, const float *gamma, const float *beta,
                       float eps, float momentum, float *running_mean, float *running_var, float *scale,
                       float *shift, float *mean, float *invstd, void *stream)
{
    if (!slab || nslab <= 0 || C <= 0 || count <= 0 || !gamma || !beta || !scale || !shift || !mean || !invstd ||
        ((running_mean == nullptr) != (running_var == nullptr)))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), slab, nslab, C, count, gamma,
                       beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
    return prifit_check_launch();
}

int prifit_gn_finalize_supported(int C, int groups)
{
    return C > 0 && groups > 0 && C % groups == 0 && C / groups <= 256 && 256 % (C / groups) == 0;
}

int prifit_gn_finalize(const float *slab, int Bs, int slabs_per_sample, int C, int groups, double count, const float *gamma,
                       const float *beta, double eps, float *scale, float *shift, float *mean, float *invstd, double *chsum,
                       void *stream)
{
    if (!slab || !gamma || !beta || !scale || !shift || !mean || !invstd || Bs <= 0 || slabs_per_sample <= 0 || count <= 0 ||
        !prifit_gn_finalize_supported(C, groups))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, Bs), dim3(256), 0, as_stream(stream), slab, slabs_per_sample, C,
                       C / groups, count, gamma, beta, eps, (const float *)nullptr, 0.0, scale, shift, mean, invstd, chsum);
    return prifit_check_launch();
}

int prifit_gn_finalize_offset(const float *slab, int Bs, int slabs_per_sample, int C, int groups, double count,
                              const float *gamma, const float *beta, double eps, const float *offset, double rows_per_sample,
                              float *scale, float *shift, float *mean, float *invstd, double *chsum, void *stream)
{
    if (!slab || !gamma || !beta || !offset || !scale || !shift || !mean || !invstd || Bs <= 0 || slabs_per_sample <= 0 ||
        count <= 0 || rows_per_sample <= 0 || !prifit_gn_finalize_supported(C, groups))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, Bs), dim3(256), 0, as_stream(stream), slab, slabs_per_sample, C,
                       C / groups, count, gamma, beta, eps, offset, rows_per_sample, scale, shift, mean, invstd, chsum);
    return prifit_check_launch();
}

int prifit_gn_bwd_finalize(const float *slab, int Bs, int slabs_per_sample, int C, int groups, double count,
                           const float *gamma, const float *mean, const float *invstd, float *coef_b, float *coef_d,
                           double *S, const double *chsum, double rows_per_sample, float *dsum, void *stream)
{
    if (!slab || !gamma || !mean || !invstd || !coef_b || !coef_d || !S || Bs <= 0 || slabs_per_sample <= 0 || count <= 0 ||
        !prifit_gn_finalize_supported(C, groups) || (chsum && (!dsum || rows_per_sample <= 0)))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups, Bs), dim3(256), 0, as_stream(stream), slab, slabs_per_sample, C,
                       C / groups, count, gamma, mean, invstd, coef_b, coef_d, S, chsum, rows_per_sample, dsum);
    return prifit_check_launch();
}

long long prifit_col_sum_workspace(int P, int C)
{
    if (P <= 0 || C <= 0) return 0;
    return (long long)((P + CS_ROWS - 1) / CS_ROWS) * C;
}

int prifit_col_sum(const float *Y, long long ld, int P, int C, float *out, float *workspace, void *stream)
{
    if (bad_mat(Y, ld, C) || !out || !workspace || ((uintptr_t)workspace & 15) || P <= 0 || C <= 0) return PRIFIT_EINVAL;
    const int nb = (P + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(col_sum_kernel, dim3(nb), dim3(256), 0, as_stream(stream), Y, ld, P, C, workspace);
    hipLaunchKernelGGL(col_sum_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), workspace, nb, C, out);
    return prifit_check_launch();
}

int prifit_col_sum_samples(const float *Y, long long ld, int P, int C, int rows_per_sample, float *out, float *workspace,
                           void *stream)
{
    if (bad_mat(Y, ld, C) || !out || !workspace || ((uintptr_t)workspace & 15) || P <= 0 || C <= 0 || rows_per_sample <= 0 ||
        rows_per_sample % CS_ROWS || P % rows_per_sample || P / rows_per_sample > 65535)
        return PRIFIT_EINVAL;
    const int nb = P / CS_ROWS, per = rows_per_sample / CS_ROWS;
    hipLaunchKernelGGL(col_sum_kernel, dim3(nb), dim3(256), 0, as_stream(stream), Y, ld, P, C, workspace);
    hipLaunchKernelGGL(col_sum_reduce_kernel, dim3((C + 255) / 256, P / rows_per_sample), dim3(256), 0, as_stream(stream), workspace,
                       per, C, out);
    return prifit_check_launch();
}

int prifit_gn_param_grads(const double *S, int Bs, int C, float *dgamma, float *dbeta, const float *dsum, float *db, void *stream)
{
    if (!S || !dgamma || !dbeta || Bs <= 0 || C <= 0 || (dsum && !db)) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(gn_param_grads_kernel, dim3(((dsum ? 3 : 2) * C + 255) / 256), dim3(256), 0, as_stream(stream), S, Bs, C,
                       dgamma, dbeta, dsum, db);
    return prifit_check_launch();
}

int prifit_affine_relu(const float *Y, long long ldy, const float *scale, const float *shift, int P, int C,
                       int rows_per_sample, float slope, float *out, long long ldo, void *stream)
{
    if (bad_mat(Y, ldy, C) || bad_mat(out, ldo, C) || !scale || !shift || P <= 0 || rows_per_sample < 0)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(affine_relu_kernel, dim3(ew_grid((long long)P * (C / 4))), dim3(256), 0, as_stream(stream),
                       Y, ldy, scale, shift, P, C / 4, rows_per_sample, slope, out, ldo);
    return prifit_check_launch();
}

int prifit_pool_fwd(const float *Y, long long ldy, const float *scale, const float *shift, int G, int K, int C,
                    int rows_per_sample, float slope, float *out, long long ldo, int32_t *arg, void *stream)
{
    if (bad_mat(Y, ldy, C) || bad_mat(out, ldo, C) || !scale || !shift || !arg || G <= 0 || K <= 0 ||
        rows_per_sample < 0 || (rows_per_sample % K) != 0)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(ew_grid((long long)G * (C / 4))), dim3(256), 0, as_stream(stream), Y,
                       ldy, scale, shift, G, K, C / 4, rows_per_sample, slope, out, ldo, arg);
    return prifit_check_launch();
}

int prifit_gather_linear_fwd(const float *U, const float *Vc, const float *bias, const int32_t *idx, int B, int N,
                             int S, int K, int C, float *Y, float *slab, void *stream)
{
    if (!U || !Vc || !idx || !Y || !slab || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0 || (C & 3) ||
        ((uintptr_t)U & 15) || ((uintptr_t)Vc & 15) || ((uintptr_t)Y & 15))
        return PRIFIT_EINVAL;
    const long long P = (long long)B * S * K;
    if (P > 2147483647LL) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(gather_linear_fwd_kernel, dim3((unsigned)((P + RED_ROWS - 1) / RED_ROWS)), dim3(256), 0,
                       as_stream(stream), U, Vc, bias, idx, N, S, K, (int)P, C, Y, slab);
    return prifit_check_launch();
}

int prifit_gather_linear_bwd(const float *dY, const int32_t *idx, int B, int N, int S, int K, int C, float *dU,
                             float *dVc, void *stream)
{
    if (!dY || !idx || !dU || !dVc || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0) return PRIFIT_EINVAL;
    const long long total = (long long)B * S * C;
    hipLaunchKernelGGL(gather_linear_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), dY, idx, N, S,
                       K, C, total, dU, dVc);
    return prifit_check_launch();
}

int prifit_gather_linear_bwd_bn_supported(int N, int C) { return (N > 0 && C > 0 && C <= GLB_CMAX && C % 2 == 0) ? 1 : 0; }

int prifit_gather_linear_bwd_bn(const float *G, const float *Y, const float *scale, const float *shift, const float *coef_a,
                                const float *coef_b, const float *coef_d, const int32_t *idx, int B, int N, int S, int K,
                                int C, float *dU, float *dVc, void *stream)
{
    if (!G || !Y || !scale || !shift || !coef_a || !coef_b || !coef_d || !idx || !dU || !dVc || B <= 0 || B > 65535 ||
        S <= 0 || K <= 0 || !prifit_gather_linear_bwd_bn_supported(N, C))
        return PRIFIT_EINVAL;
    // centre ranges per (shape, chunk): enough workgroups for two per CU, at least 16 centres each
    const int ranges = (N + GLB_PTS - 1) / GLB_PTS;
    int splits = (2 * 256) / (B * ranges);                 // one round of resident workgroups (two per CU), no tail
    splits = splits < 1 ? 1 : (splits > (S + 7) / 8 ? (S + 7) / 8 : splits);
    hipLaunchKernelGGL(gather_linear_bwd_bn_kernel, dim3(ranges, splits, B), dim3(GLB_NTH), 0, as_stream(stream), G, Y, scale,
                       shift, coef_a, coef_b, coef_d, idx, N, S, K, C, splits, dU, dVc);
    return prifit_check_launch();
}

int prifit_bn_tail_replicas(void) { return BN_TAIL_REPLICAS; }

int prifit_bn_relu_bwd_reduce(const float *G, long long ldg, const float *Y, long long ldy, const float *scale,
                              const float *shift, const float *mean, const float *invstd, int P, int C,
                              int rows_per_sample, float slope, float *slab, const prifit_bn_bwd *bn, void *stream)
{
    const bool tail = bn && bn->acc;
    if (bad_mat(G, ldg, C) || bad_mat(Y, ldy, C) || !scale || !shift || !mean || !invstd || (!slab && !tail) || P <= 0 ||
        rows_per_sample < 0 || (rows_per_sample % RED_ROWS) != 0 || bn_bwd_bad(bn) || (tail && rows_per_sample != 0))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3((P + RED_ROWS - 1) / RED_ROWS), dim3(256), 0,
                       as_stream(stream), G, ldg, Y, ldy, scale, shift, mean, invstd, P, C, rows_per_sample, slope,
                       slab, bn_tail_bwd(bn, C));
    return prifit_check_launch();
}

int prifit_pool_bwd_reduce(const float *gp, long long ldgp, const float *Y, long long ldy, const int32_t *arg,
                           const float *scale, const float *shift, const float *mean, const float *invstd, int G,
                           int K, int C, int rows_per_sample, float slope, float *slab, const prifit_bn_bwd *bn, void *stream)
{
    const bool tail = bn && bn->acc;
    if (bad_mat(gp, ldgp, C) || bad_mat(Y, ldy, C) || !arg || !scale || !shift || !mean || !invstd || (!slab && !tail) ||
        G <= 0 || K <= 0 || rows_per_sample < 0 || (rows_per_sample % (K * POOL_RED_ROWS)) != 0 || bn_bwd_bad(bn) ||
        (tail && rows_per_sample != 0))
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(pool_bwd_reduce_kernel, dim3((G + POOL_RED_ROWS - 1) / POOL_RED_ROWS), dim3(256), 0,
                       as_stream(stream), gp, ldgp, Y, ldy, arg, scale, shift, mean, invstd, G, K, C, rows_per_sample,
                       slope, slab, bn_tail_bwd(bn, C));
    return prifit_check_launch();
}

int prifit_bn_bwd_finalize(const float *slab, int nslab, int C, double count, int training, const float *scale,
                           const float *mean, const float *invstd, float *dgamma, float *dbeta, float *coef_a,
                           float *coef_b, float *coef_d, void *stream)
{
    if (!slab || nslab <= 0 || C <= 0 || count <= 0 || !scale || !mean || !invstd || !dgamma || !dbeta ||
        !coef_a || !coef_b || !coef_d)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), slab, nslab, C, count,
                       training, scale, mean, invstd, dgamma, dbeta, coef_a, coef_b, coef_d);
    return prifit_check_launch();
}

int prifit_bn_relu_bwd_apply(const float *G, long long ldg, const float *Y, long long ldy, const float *scale,
                             const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                             int P, int C, int rows_per_sample, float slope, float *dY, long long ldd, void *stream)
{
    if (bad_mat(G, ldg, C) || bad_mat(Y, ldy, C) || bad_mat(dY, ldd, C) || !scale || !shift || !coef_a ||
        !coef_b || !coef_d || P <= 0 || rows_per_sample < 0)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(ew_grid((long long)P * (C / 4))), dim3(256), 0,
                       as_stream(stream), G, ldg, Y, ldy, scale, shift, coef_a, coef_b, coef_d, P, C / 4,
                       rows_per_sample, slope, dY, ldd);
    return prifit_check_launch();
}

int prifit_pool_bwd_apply(const float *gp, long long ldgp, const float *Y, long long ldy, const int32_t *arg,
                          const float *scale, const float *shift, const float *coef_a, const float *coef_b,
                          const float *coef_d, int G, int K, int C, int rows_per_sample, float slope, float *dY,
                          long long ldd, void *stream)
{
    if (bad_mat(gp, ldgp, C) || bad_mat(Y, ldy, C) || bad_mat(dY, ldd, C) || !arg || !scale || !shift ||
        !coef_a || !coef_b || !coef_d || G <= 0 || K <= 0 || rows_per_sample < 0)
        return PRIFIT_EINVAL;
    const int C4 = C / 4;
    // the row-walking form takes the per-sample coefficient row from a group's FIRST row: groups must not straddle samples
    // (and a thread walks ALL K rows of its group: only for the usual short groups -- a whole-cloud pool, K = N, takes the
    // element-per-thread form)
    if (C4 <= 256 && 256 % C4 == 0 && K <= 256 && (long long)G * K < 2147483647LL &&
        (rows_per_sample == 0 || rows_per_sample % K == 0))
        hipLaunchKernelGGL(pool_bwd_apply_rows_kernel, dim3(ew_grid((long long)G * C4)), dim3(256), 0, as_stream(stream),
                           gp, ldgp, Y, ldy, arg, scale, shift, coef_a, coef_b, coef_d, G, K, C4, rows_per_sample, slope, dY,
                           ldd);
    else
        hipLaunchKernelGGL(pool_bwd_apply_kernel, dim3(ew_grid((long long)G * K * C4)), dim3(256), 0,
                           as_stream(stream), gp, ldgp, Y, ldy, arg, scale, shift, coef_a, coef_b, coef_d, G, K, C4,
                           rows_per_sample, slope, dY, ldd);
    return prifit_check_launch();
}

int prifit_gather_linear_bwd_pool(const float *gp, long long ldgp, const float *Y, const int32_t *arg, const float *scale,
                                  const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                                  const int32_t *idx, int B, int N, int S, int K, int C, int rows_per_sample, float slope,
                                  float *dU, float *dVc, void *stream)
{
    if (!gp || !Y || !arg || !scale || !shift || !coef_a || !coef_b || !coef_d || !idx || !dU || !dVc || B <= 0 || N <= 0 ||
        S <= 0 || K <= 0 || C <= 0 || ldgp < C || rows_per_sample < 0 || (rows_per_sample % K) != 0)
        return PRIFIT_EINVAL;
    const long long total = (long long)B * S * C;
    hipLaunchKernelGGL(gather_linear_bwd_pool_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), gp, ldgp, Y, arg,
                       scale, shift, coef_a, coef_b, coef_d, idx, N, S, K, C, rows_per_sample, slope, total, dU, dVc);
    return prifit_check_launch();
}

int prifit_pool_bwd_table(const float *gp, long long ldgp, const float *Y, long long ldy, const int32_t *arg,
                          const float *scale, const float *shift, const float *coef_a, int G, int K, int C, float slope,
                          float *T, void *stream)
{
    if (!gp || !Y || !arg || !scale || !shift || !coef_a || !T || G <= 0 || K <= 0 || C <= 0 || ldgp < C || ldy < C)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(pool_bwd_table_kernel, dim3(ew_grid((long long)G * C)), dim3(256), 0, as_stream(stream), gp, ldgp,
                       Y, ldy, arg, scale, shift, coef_a, G, K, C, slope, T);
    return prifit_check_launch();
}

int prifit_pool_from_candidates(const float *cand, const float *scale, const float *shift, int G, int K, int C,
                                int rows_per_sample, float slope, float *out, long long ldo, int32_t *arg, float *ystar,
                                void *stream)
{
    if (!cand || !scale || !shift || !out || !arg || G <= 0 || K < 32 || (K & 31) || C <= 0 || ldo < C || rows_per_sample < 0 ||
        (rows_per_sample % K) != 0)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(pool_from_candidates_kernel, dim3(ew_grid((long long)G * C)), dim3(256), 0, as_stream(stream), cand,
                       scale, shift, G, K, C, rows_per_sample, slope, out, ldo, arg, ystar);
    return prifit_check_launch();
}

}  // extern "C"
