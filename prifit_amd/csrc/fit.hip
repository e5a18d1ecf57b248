// Weighted ellipsoid fitting and the analytic-chamfer loss terms (src/ellipsoid_fitting.py,
// src/fitting_utils.py, src/ellipsoid_utils.py, src/sample_ellipsoid.py, convex_loss.py, src/utils.py).
//
// Fixed-capacity layout: every shape owns KM cluster slots; slot k of shape b is live when
// k < count[b] and valid[b][k] != 0 (ill-conditioned fits are dropped by a validity flag instead of
// the reference's -1 sentinel).  One workgroup per (shape, cluster) for the fit: the ten weighted
// moments (sum w, sum w p, sum w q q^T) are block reductions, the 3x3 SVD and its custom backward run
// in one lane in double precision, the extent pass is a block (value, index) max/min reduction.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// block reductions (256 threads)
// ---------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float *s_buf /* [4*NV] */)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_f32(v[i]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) s_buf[(threadIdx.x >> 6) * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (s_buf[i] + s_buf[NV + i]) + (s_buf[2 * NV + i] + s_buf[3 * NV + i]);
}

// ---------------------------------------------------------------------------------------------
// 3x3 SVD in double (Jacobi eigen-decomposition of M^T M), singular values descending.
// ---------------------------------------------------------------------------------------------
__device__ void svd3(const double M[3][3], double U[3][3], double S[3], double V[3][3])
{
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += M[k][i] * M[k][j];
            A[i][j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    double lam[3] = {A[0][0], A[1][1], A[2][2]};
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (lam[ord[j]] < lam[ord[j + 1]]) { int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double Vs[3][3];
    for (int j = 0; j < 3; ++j) {
        S[j] = sqrt(lam[ord[j]] > 0 ? lam[ord[j]] : 0.0);
        for (int i = 0; i < 3; ++i) Vs[i][j] = V[i][ord[j]];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
    for (int j = 0; j < 3; ++j) {
        const double inv = S[j] > 1e-300 ? 1.0 / S[j] : 0.0;
        for (int i = 0; i < 3; ++i) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += M[i][k] * V[k][j];
            U[i][j] = s * inv;
        }
    }
}

// saved-state layout per (b, k): 48 floats
//  [0..2] centre, [3] sw, [4..12] cov (row-major 3x3), [13..21] U, [22..24] S, [25..33] V (canonical, before
//  the determinant flip), [34] flip (0/1), [35..40] extreme indices as float bits (imax0..2, imin0..2)
constexpr int FIT_STATE = 48;

__global__ __launch_bounds__(256) void ellipsoid_fit_fwd_kernel(
    const float *__restrict__ pts, const float *__restrict__ W, const int32_t *__restrict__ count,
    const float *__restrict__ rnd, long long rnd_stride_b, long long rnd_stride_k, int canonical, int N, int KM,
    float *__restrict__ r_o, float *__restrict__ V_o, float *__restrict__ c_o, int32_t *__restrict__ valid_o,
    float *__restrict__ state)
{
    __shared__ float s_buf[4 * 6];
    __shared__ float s_par[16];
    __shared__ unsigned long long s_ext[4 * 6];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t slot = (size_t)b * KM + k;
    if (k >= count[b]) {
        if (tid == 0) valid_o[slot] = 0;
        if (tid < 3) { r_o[slot * 3 + tid] = 0.f; c_o[slot * 3 + tid] = 0.f; }
        if (tid < 9) V_o[slot * 9 + tid] = (tid % 4 == 0) ? 1.f : 0.f;
        return;
    }
    const float *P = pts + (size_t)b * N * 3;
    const float *Wk = W + (size_t)b * N * KM + k;
    // pass 1: sum w, sum w p
    float a1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < N; i += 256) {
        const float w = Wk[(size_t)i * KM];
        a1[0] += w; a1[1] += w * P[i * 3]; a1[2] += w * P[i * 3 + 1]; a1[3] += w * P[i * 3 + 2];
    }
    block_sum<4>(a1, s_buf);
    const float sw = a1[0];
    const float cx = a1[1] / sw, cy = a1[2] / sw, cz = a1[3] / sw;
    // pass 2: weighted second moments about the centre
    float a2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < N; i += 256) {
        const float w = Wk[(size_t)i * KM];
        const float x = P[i * 3] - cx, y = P[i * 3 + 1] - cy, z = P[i * 3 + 2] - cz;
        a2[0] += w * x * x; a2[1] += w * x * y; a2[2] += w * x * z;
        a2[3] += w * y * y; a2[4] += w * y * z; a2[5] += w * z * z;
    }
    block_sum<6>(a2, s_buf);
    float *st = state + slot * FIT_STATE;
    if (tid == 0) {
        const float cov[3][3] = {{a2[0] / sw, a2[1] / sw, a2[2] / sw},
                                 {a2[1] / sw, a2[3] / sw, a2[4] / sw},
                                 {a2[2] / sw, a2[4] / sw, a2[5] / sw}};
        float mean = 0.f;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) mean += cov[i][j];
        mean /= 9.f;
        const float *R = rnd + b * rnd_stride_b + k * rnd_stride_k;
        double M[3][3], U[3][3], S[3], V[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[i][j] = (double)(cov[i][j] + 1e-4f * mean * R[i * 3 + j]);
        svd3(M, U, S, V);
        const int ok = !((float)S[0] / (float)S[2] > 1e5f);  // src/ellipsoid_fitting.py:43
        if (canonical)
            for (int j = 0; j < 3; ++j) {
                int im = 0;
                for (int i = 1; i < 3; ++i)
                    if (fabs(V[i][j]) > fabs(V[im][j])) im = i;
                if (V[im][j] < 0)
                    for (int i = 0; i < 3; ++i) { V[i][j] = -V[i][j]; U[i][j] = -U[i][j]; }
            }
        const double det = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) -
                           V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                           V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
        const int flip = det < 0;
        st[0] = cx; st[1] = cy; st[2] = cz; st[3] = sw;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                st[4 + i * 3 + j] = cov[i][j];
                st[13 + i * 3 + j] = (float)U[i][j];
                st[25 + i * 3 + j] = (float)V[i][j];
                const float vp = (float)((j == 2 && flip) ? -V[i][j] : V[i][j]);
                s_par[i * 3 + j] = vp;
                V_o[slot * 9 + i * 3 + j] = vp;
            }
        for (int j = 0; j < 3; ++j) st[22 + j] = (float)S[j];
        st[34] = (float)flip;
        valid_o[slot] = ok;
        c_o[slot * 3] = cx; c_o[slot * 3 + 1] = cy; c_o[slot * 3 + 2] = cz;
    }
    __syncthreads();
    // pass 3: extents of the weight-scaled points along the principal axes (principal_axis_ellipsoid, "slow")
    float vp[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) vp[i] = s_par[i];
    // keys: (order-preserving float image << 32) | ~index  -> max key = first maximum; for the minimum the
    // float image is inverted
    unsigned long long kmax[3] = {0ull, 0ull, 0ull}, kmin[3] = {0ull, 0ull, 0ull};
    for (int i = tid; i < N; i += 256) {
        const float w = Wk[(size_t)i * KM];
        const float x = (P[i * 3] - cx) * w, y = (P[i * 3 + 1] - cy) * w, z = (P[i * 3 + 2] - cz) * w;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = x * vp[a] + y * vp[3 + a] + z * vp[6 + a];
            unsigned u = __float_as_uint(t);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            const unsigned long long lo = (unsigned)(0xffffffffu - (unsigned)i);
            const unsigned long long k1 = ((unsigned long long)u << 32) | lo;
            const unsigned long long k2 = ((unsigned long long)(~u) << 32) | lo;
            kmax[a] = k1 > kmax[a] ? k1 : kmax[a];
            kmin[a] = k2 > kmin[a] ? k2 : kmin[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { kmax[a] = wave_max_u64(kmax[a]); kmin[a] = wave_max_u64(kmin[a]); }
    if ((tid & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_ext[(tid >> 6) * 6 + a] = kmax[a]; s_ext[(tid >> 6) * 6 + 3 + a] = kmin[a]; }
    __syncthreads();
    if (tid < 3) {
        unsigned long long m1 = 0ull, m2 = 0ull;
        for (int w = 0; w < 4; ++w) {
            m1 = s_ext[w * 6 + tid] > m1 ? s_ext[w * 6 + tid] : m1;
            m2 = s_ext[w * 6 + 3 + tid] > m2 ? s_ext[w * 6 + 3 + tid] : m2;
        }
        unsigned u1 = (unsigned)(m1 >> 32), u2 = ~(unsigned)(m2 >> 32);
        u1 = (u1 & 0x80000000u) ? (u1 & 0x7fffffffu) : ~u1;
        u2 = (u2 & 0x80000000u) ? (u2 & 0x7fffffffu) : ~u2;
        const float mx = __uint_as_float(u1), mn = __uint_as_float(u2);
        r_o[slot * 3 + tid] = fabsf(mx - mn) / 2.0f;
        st[35 + tid] = __int_as_float((int)(0xffffffffu - (unsigned)(m1 & 0xffffffffu)));
        st[38 + tid] = __int_as_float((int)(0xffffffffu - (unsigned)(m2 & 0xffffffffu)));
    }
}

// Backward: (g_r, g_Vp, g_c) -> dL/dW[:, k].
__global__ __launch_bounds__(256) void ellipsoid_fit_bwd_kernel(
    const float *__restrict__ pts, const float *__restrict__ W, const int32_t *__restrict__ count,
    const int32_t *__restrict__ valid, const float *__restrict__ rnd, long long rnd_stride_b,
    long long rnd_stride_k, const float *__restrict__ state, const float *__restrict__ g_r,
    const float *__restrict__ g_V, const float *__restrict__ g_c, int N, int KM, float *__restrict__ gW)
{
    __shared__ float s_g[9 + 1 + 3 + 6];  // g_cov, tr(g_cov^T cov), g_c_total, sparse g_w
    __shared__ int s_idx[6];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t slot = (size_t)b * KM + k;
    float *gWk = gW + (size_t)b * N * KM + k;
    if (k >= count[b] || !valid[slot]) {
        for (int i = tid; i < N; i += 256) gWk[(size_t)i * KM] = 0.f;
        return;
    }
    const float *P = pts + (size_t)b * N * 3;
    const float *Wk = W + (size_t)b * N * KM + k;
    const float *st = state + slot * FIT_STATE;
    const float cx = st[0], cy = st[1], cz = st[2], sw = st[3];
    if (tid == 0) {
        double U[3][3], S[3], V[3][3], Vp[3][3], cov[3][3], gVp[3][3], gcen[3], gw_sp[6];
        const int flip = st[34] != 0.f;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                cov[i][j] = st[4 + i * 3 + j];
                U[i][j] = st[13 + i * 3 + j];
                V[i][j] = st[25 + i * 3 + j];
                Vp[i][j] = (j == 2 && flip) ? -V[i][j] : V[i][j];
                gVp[i][j] = g_V[slot * 9 + i * 3 + j];
            }
        for (int j = 0; j < 3; ++j) { S[j] = st[22 + j]; gcen[j] = g_c[slot * 3 + j]; }
        // extents: r_a = |max_a - min_a| / 2, gradient to the two extreme rows of each axis
        for (int a = 0; a < 3; ++a) {
            const int im[2] = {__float_as_int(st[35 + a]), __float_as_int(st[38 + a])};
            double tv[2], q[2][3], ww[2];
            for (int e = 0; e < 2; ++e) {
                const int i = im[e];
                ww[e] = Wk[(size_t)i * KM];
                q[e][0] = (double)P[i * 3] - cx; q[e][1] = (double)P[i * 3 + 1] - cy; q[e][2] = (double)P[i * 3 + 2] - cz;
                tv[e] = ww[e] * (q[e][0] * Vp[0][a] + q[e][1] * Vp[1][a] + q[e][2] * Vp[2][a]);
            }
            const double sgn = (tv[0] - tv[1]) > 0 ? 1.0 : ((tv[0] - tv[1]) < 0 ? -1.0 : 0.0);
            for (int e = 0; e < 2; ++e) {
                const double coef = (e == 0 ? 1.0 : -1.0) * sgn * (double)g_r[slot * 3 + a] / 2.0;
                const double proj = q[e][0] * Vp[0][a] + q[e][1] * Vp[1][a] + q[e][2] * Vp[2][a];
                gw_sp[e * 3 + a] = coef * proj;
                for (int d = 0; d < 3; ++d) {
                    gcen[d] -= coef * ww[e] * Vp[d][a];      // q = p - c
                    gVp[d][a] += coef * ww[e] * q[e][d];
                }
                s_idx[e * 3 + a] = im[e];
            }
        }
        // through the determinant flip back to the SVD's V
        double gV[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) gV[i][j] = (j == 2 && flip) ? -gVp[i][j] : gVp[i][j];
        // CustomSVD backward (src/fitting_utils.py:67-105) with dL/dS = 0, dL/dU ignored
        double Kc[3][3], A1[3][3], sym[3][3], T[3][3], Gm[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                if (i == j) { Kc[i][j] = 0.0; continue; }
                const double diff = S[i] - S[j];
                const double sg = diff > 0 ? 1.0 : (diff < 0 ? -1.0 : 0.0);
                const double kneg = sg * fmax(fabs(diff), 1e-6);
                Kc[i][j] = (1.0 / kneg) * (1.0 / (S[i] + S[j]));
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int d = 0; d < 3; ++d) s += V[d][i] * gV[d][j];
                A1[i][j] = Kc[j][i] * s;  // K^T * (V^T gV)
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) sym[i][j] = (A1[i][j] + A1[j][i]) / 2.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int d = 0; d < 3; ++d) s += U[i][d] * S[d] * sym[d][j];
                T[i][j] = s;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int d = 0; d < 3; ++d) s += T[i][d] * V[j][d];
                Gm[i][j] = 2.0 * s;
            }
        // M = cov + 1e-4 * mean(cov) * R
        const float *R = rnd + b * rnd_stride_b + k * rnd_stride_k;
        double gR = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) gR += Gm[i][j] * (double)R[i * 3 + j];
        double tr = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double gc = Gm[i][j] + 1e-4 * gR / 9.0;
                s_g[i * 3 + j] = (float)gc;
                tr += gc * cov[i][j];
            }
        s_g[9] = (float)tr;
        for (int d = 0; d < 3; ++d) s_g[10 + d] = (float)gcen[d];
        for (int e = 0; e < 6; ++e) s_g[13 + e] = (float)gw_sp[e];
    }
    __syncthreads();
    float gc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gc[i] = s_g[i];
    const float tr = s_g[9], gx = s_g[10], gy = s_g[11], gz = s_g[12];
    for (int i = tid; i < N; i += 256) {
        const float x = P[i * 3] - cx, y = P[i * 3 + 1] - cy, z = P[i * 3 + 2] - cz;
        const float quad = x * (gc[0] * x + gc[1] * y + gc[2] * z) + y * (gc[3] * x + gc[4] * y + gc[5] * z) +
                           z * (gc[6] * x + gc[7] * y + gc[8] * z);
        float v = (quad - tr) / sw + (gx * x + gy * y + gz * z) / sw;
#pragma unroll
        for (int e = 0; e < 6; ++e)
            if (s_idx[e] == i) v += s_g[13 + e];
        gWk[(size_t)i * KM] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// ellipsoid SDF of the target points, min over ellipsoids of |sdf|, squared (convex_loss.py:313-328,
// src/utils.py:410-411).  One thread per target point, parameters in LDS.
// ---------------------------------------------------------------------------------------------
struct EllParam { float r[3]; float V[9]; float c[3]; };

__device__ __forceinline__ float sdf_eval(const EllParam &e, float px, float py, float pz, float q[3], float &k0,
                                          float &k1)
{
    const float dx = px - e.c[0], dy = py - e.c[1], dz = pz - e.c[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = e.V[a] * dx + e.V[3 + a] * dy + e.V[6 + a] * dz;  // V^T (p - c)
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = q[a] / (e.r[a] + 1e-6f), v = q[a] / (e.r[a] * e.r[a] + 1e-6f);
        s0 += u * u;
        s1 += v * v;
    }
    k0 = sqrtf(s0);
    k1 = sqrtf(s1);
    return k0 * (k0 - 1.0f) / (k1 + 1e-6f);
}

// Cuboid with half-sides r in the same frame (convex_loss.py:473-487): q = |V^T (p - c)| - r,
// sdf = ||relu(q)|| + min(max(q), 0).  `q` returns the shifted point as for the ellipsoid.
__device__ __forceinline__ float sdf_eval_cuboid(const EllParam &e, float px, float py, float pz, float q[3])
{
    const float dx = px - e.c[0], dy = py - e.c[1], dz = pz - e.c[2];
    float ss = 0.f, mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q[a] = e.V[a] * dx + e.V[3 + a] * dy + e.V[6 + a] * dz;
        const float t = fabsf(q[a]) - e.r[a];
        const float rl = fmaxf(t, 0.f);
        ss += rl * rl;
        mx = fmaxf(mx, t);
    }
    return sqrtf(ss) + fminf(mx, 0.f);
}

constexpr int KIND_ELLIPSOID = 0, KIND_CUBOID = 1;

__device__ __forceinline__ float prim_eval(int kind, const EllParam &e, float px, float py, float pz, float q[3],
                                           float &k0, float &k1)
{
    if (kind == KIND_CUBOID) { k0 = k1 = 0.f; return sdf_eval_cuboid(e, px, py, pz, q); }
    return sdf_eval(e, px, py, pz, q, k0, k1);
}

// gf = dL/d(sdf)  ->  gq = dL/d(shifted point), gr = dL/d(r)
__device__ __forceinline__ void prim_grad(int kind, const EllParam &e, const float q[3], float k0, float k1, float gf,
                                          float gq[3], float gr[3])
{
    if (kind == KIND_CUBOID) {
        float t[3], ss = 0.f, mx = -INFINITY;
        int am = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            t[a] = fabsf(q[a]) - e.r[a];
            const float rl = fmaxf(t[a], 0.f);
            ss += rl * rl;
            if (t[a] > mx) { mx = t[a]; am = a; }   // first maximum, as torch.max
        }
        const float nrm = sqrtf(ss);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float d = (t[a] > 0.f && nrm > 0.f) ? t[a] / nrm : 0.f;
            if (a == am && mx <= 0.f) d += 1.0f;     // clamp_max(max(q), 0) passes the gradient while max <= 0
            const float sg = q[a] > 0.f ? 1.0f : (q[a] < 0.f ? -1.0f : 0.f);
            gq[a] = gf * d * sg;
            gr[a] = -gf * d;
        }
        return;
    }
    const float den = k1 + 1e-6f;
    const float df0 = (2.0f * k0 - 1.0f) / den, df1 = -k0 * (k0 - 1.0f) / (den * den);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ra = e.r[a] + 1e-6f, rb = e.r[a] * e.r[a] + 1e-6f;
        const float u = q[a] / ra, v = q[a] / rb;
        const float dk0dq = k0 > 0.f ? u / (k0 * ra) : 0.f, dk1dq = k1 > 0.f ? v / (k1 * rb) : 0.f;
        const float dk0dr = k0 > 0.f ? -u * u / (k0 * ra) : 0.f;
        const float dk1dr = k1 > 0.f ? -2.0f * e.r[a] * v * v / (k1 * rb) : 0.f;
        gq[a] = gf * (df0 * dk0dq + df1 * dk1dq);
        gr[a] = gf * (df0 * dk0dr + df1 * dk1dr);
    }
}

constexpr int KM_MAX = 64;

__global__ __launch_bounds__(256) void sdf_fwd_kernel(int kind, const float *__restrict__ tgt, int M,
                                                      const float *__restrict__ r, const float *__restrict__ V,
                                                      const float *__restrict__ c,
                                                      const int32_t *__restrict__ valid, int KM,
                                                      int32_t *__restrict__ arg, float *__restrict__ fval,
                                                      float *__restrict__ sum_o)
{
    __shared__ EllParam s_e[KM_MAX];
    __shared__ int s_ok[KM_MAX];
    __shared__ float s_red[4];
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < KM; k += 256) {
        const size_t s = (size_t)b * KM + k;
        s_ok[k] = valid[s];
        for (int i = 0; i < 3; ++i) { s_e[k].r[i] = r[s * 3 + i]; s_e[k].c[i] = c[s * 3 + i]; }
        for (int i = 0; i < 9; ++i) s_e[k].V[i] = V[s * 9 + i];
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    float best = INFINITY, bf = 0.f;
    int bk = -1;
    if (m < M) {
        const float *p = tgt + ((size_t)b * M + m) * 3;
        const float px = p[0], py = p[1], pz = p[2];
        for (int k = 0; k < KM; ++k) {
            if (!s_ok[k]) continue;
            float q[3], k0, k1;
            const float f = prim_eval(kind, s_e[k], px, py, pz, q, k0, k1);
            if (fabsf(f) < best) { best = fabsf(f); bf = f; bk = k; }
        }
        arg[(size_t)b * M + m] = bk;
        fval[(size_t)b * M + m] = bf;
    }
    float v = (m < M && bk >= 0) ? bf * bf : 0.f;
    v = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sum_o + b, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

// d(sum f^2)/d(r, V, c), scaled per shape by gscale[b]
__global__ __launch_bounds__(256) void sdf_bwd_kernel(int kind, const float *__restrict__ tgt, int M,
                                                      const float *__restrict__ r, const float *__restrict__ V,
                                                      const float *__restrict__ c,
                                                      const int32_t *__restrict__ arg,
                                                      const float *__restrict__ gscale, int KM,
                                                      float *__restrict__ g_r, float *__restrict__ g_V,
                                                      float *__restrict__ g_c)
{
    __shared__ EllParam s_e[KM_MAX];
    __shared__ float s_acc[KM_MAX * 15];
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < KM; k += 256) {
        const size_t s = (size_t)b * KM + k;
        for (int i = 0; i < 3; ++i) { s_e[k].r[i] = r[s * 3 + i]; s_e[k].c[i] = c[s * 3 + i]; }
        for (int i = 0; i < 9; ++i) s_e[k].V[i] = V[s * 9 + i];
    }
    for (int i = threadIdx.x; i < KM * 15; i += 256) s_acc[i] = 0.f;
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < M) {
        const int k = arg[(size_t)b * M + m];
        if (k >= 0) {
            const EllParam &e = s_e[k];
            const float *p = tgt + ((size_t)b * M + m) * 3;
            float q[3], k0, k1;
            const float f = prim_eval(kind, e, p[0], p[1], p[2], q, k0, k1);
            const float gf = gscale[b] * 2.0f * f;
            float gq[3], gr[3];
            float *acc = s_acc + k * 15;
            prim_grad(kind, e, q, k0, k1, gf, gq, gr);
#pragma unroll
            for (int a = 0; a < 3; ++a) atomicAdd(acc + a, gr[a]);
            const float d[3] = {p[0] - e.c[0], p[1] - e.c[1], p[2] - e.c[2]};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float gc = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    atomicAdd(acc + 3 + i * 3 + a, d[i] * gq[a]);  // q_a = sum_i V[i][a] d_i
                    gc -= e.V[i * 3 + a] * gq[a];
                }
                atomicAdd(acc + 12 + i, gc);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KM * 15; i += 256) {
        const float v = s_acc[i];
        if (v == 0.f) continue;
        const int k = i / 15, j = i % 15;
        const size_t s = (size_t)b * KM + k;
        if (j < 3) unsafeAtomicAdd(g_r + s * 3 + j, v);
        else if (j < 12) unsafeAtomicAdd(g_V + s * 9 + (j - 3), v);
        else unsafeAtomicAdd(g_c + s * 3 + (j - 12), v);
    }
}

// Full SDF matrix sdf[b][m][k] (0 for dead slots) and its autograd: used by the optional intersection term
// (convex_loss.py:374-413), which needs every ellipsoid's value at every point, not only the closest one.
__global__ __launch_bounds__(256) void sdf_matrix_fwd_kernel(int kind, const float *__restrict__ pts, int M,
                                                             const float *__restrict__ r,
                                                             const float *__restrict__ V,
                                                             const float *__restrict__ c,
                                                             const int32_t *__restrict__ valid, int KM,
                                                             float *__restrict__ out)
{
    __shared__ EllParam s_e[KM_MAX];
    __shared__ int s_ok[KM_MAX];
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < KM; k += 256) {
        const size_t s = (size_t)b * KM + k;
        s_ok[k] = valid[s];
        for (int i = 0; i < 3; ++i) { s_e[k].r[i] = r[s * 3 + i]; s_e[k].c[i] = c[s * 3 + i]; }
        for (int i = 0; i < 9; ++i) s_e[k].V[i] = V[s * 9 + i];
    }
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float *p = pts + ((size_t)b * M + m) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float *o = out + ((size_t)b * M + m) * KM;
    for (int k = 0; k < KM; ++k) {
        float q[3], k0, k1;
        o[k] = s_ok[k] ? prim_eval(kind, s_e[k], px, py, pz, q, k0, k1) : 0.f;
    }
}

__global__ __launch_bounds__(256) void sdf_matrix_bwd_kernel(int kind, const float *__restrict__ pts, int M,
                                                             const float *__restrict__ r,
                                                             const float *__restrict__ V,
                                                             const float *__restrict__ c,
                                                             const int32_t *__restrict__ valid,
                                                             const float *__restrict__ g, int KM,
                                                             float *__restrict__ g_r, float *__restrict__ g_V,
                                                             float *__restrict__ g_c)
{
    __shared__ EllParam s_e[KM_MAX];
    __shared__ int s_ok[KM_MAX];
    __shared__ float s_acc[KM_MAX * 15];
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < KM; k += 256) {
        const size_t s = (size_t)b * KM + k;
        s_ok[k] = valid[s];
        for (int i = 0; i < 3; ++i) { s_e[k].r[i] = r[s * 3 + i]; s_e[k].c[i] = c[s * 3 + i]; }
        for (int i = 0; i < 9; ++i) s_e[k].V[i] = V[s * 9 + i];
    }
    for (int i = threadIdx.x; i < KM * 15; i += 256) s_acc[i] = 0.f;
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < M) {
        const float *p = pts + ((size_t)b * M + m) * 3;
        for (int k = 0; k < KM; ++k) {
            const float gf = g[((size_t)b * M + m) * KM + k];
            if (!s_ok[k] || gf == 0.f) continue;
            const EllParam &e = s_e[k];
            float q[3], k0, k1;
            prim_eval(kind, e, p[0], p[1], p[2], q, k0, k1);
            float gq[3], gr[3];
            float *acc = s_acc + k * 15;
            prim_grad(kind, e, q, k0, k1, gf, gq, gr);
#pragma unroll
            for (int a = 0; a < 3; ++a) atomicAdd(acc + a, gr[a]);
            const float d[3] = {p[0] - e.c[0], p[1] - e.c[1], p[2] - e.c[2]};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float gc = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    atomicAdd(acc + 3 + i * 3 + a, d[i] * gq[a]);
                    gc -= e.V[i * 3 + a] * gq[a];
                }
                atomicAdd(acc + 12 + i, gc);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KM * 15; i += 256) {
        const float v = s_acc[i];
        if (v == 0.f) continue;
        const int k = i / 15, j = i % 15;
        const size_t s = (size_t)b * KM + k;
        if (j < 3) unsafeAtomicAdd(g_r + s * 3 + j, v);
        else if (j < 12) unsafeAtomicAdd(g_V + s * 9 + (j - 3), v);
        else unsafeAtomicAdd(g_c + s * 3 + (j - 12), v);
    }
}

// ---------------------------------------------------------------------------------------------
// surface sampling budget (src/ellipsoid_utils.py:87-107, :157-159)
// ---------------------------------------------------------------------------------------------
__global__ void sample_budget_kernel(int kind, const float *__restrict__ r, const int32_t *__restrict__ valid, int KM,
                                     int cap, int32_t *__restrict__ n_o, int32_t *__restrict__ off_o)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    double area[KM_MAX];
    double total = 0.0;
    for (int k = 0; k < KM; ++k) {
        area[k] = 0.0;
        if (!valid[(size_t)b * KM + k]) continue;
        const float a = r[((size_t)b * KM + k) * 3], bb = r[((size_t)b * KM + k) * 3 + 1],
                    cc = r[((size_t)b * KM + k) * 3 + 2];
        if (kind == KIND_CUBOID) {
            // sides 2a, 2b, 2c: area = 8 (ab + bc + ca) in fp32 as upstream (src/ellipsoid_utils.py:186-188)
            area[k] = (double)__fmul_rn(8.0f, __fadd_rn(__fadd_rn(__fmul_rn(a, bb), __fmul_rn(bb, cc)), __fmul_rn(cc, a)));
        } else {
            const float p = 1.585f;
            const float s = powf(a * bb, p) + powf(bb * cc, p) + powf(cc * a, p);
            area[k] = (double)(4.0f * 3.142f * powf(s, 1.0f / p));
        }
        total += area[k];
    }
    int off = 0;
    for (int k = 0; k < KM; ++k) {
        int n = 0;
        if (valid[(size_t)b * KM + k]) {
            n = (int)rint(10000.0 * (area[k] / total));  // np.round: half to even
            if (n <= 0) n = 100;
            if (off + n > cap) n = cap - off;
        }
        n_o[(size_t)b * KM + k] = n;
        off_o[(size_t)b * (KM + 1) + k] = off;
        off += n;
    }
    off_o[(size_t)b * (KM + 1) + KM] = off;
}

__device__ __forceinline__ void fib_dir(int j, int n, float &cu, float &su, float &cv, float &sv)
{
    const double z = 1.0 - (2.0 * (double)j + 1.0) / (double)n;
    double ip;
    const double lon = 6.283185307179586 * modf((double)j * 0.6180339887498949, &ip);
    cu = (float)cos(lon); su = (float)sin(lon);
    cv = (float)z; sv = (float)sqrt(fmax(0.0, 1.0 - z * z));
}

// The build's deterministic surface parameters of a box with half-sides (a, b, c) (replaces
// trimesh.sample.sample_surface_even of src/sample_ellipsoid.py:78-84): sample j of n goes to the face whose slice
// of the cumulative area [+z, -z, +x, -x, +y, -y] contains (j + 0.5) / n; inside the face the two free coordinates
// follow the R2 low-discrepancy sequence.  u is then scaled as upstream (:88): (u s) / (s + 1e-6), cast to fp32.
__device__ __forceinline__ void cuboid_unit(int j, int n, const float r[3], float u[3])
{
    const double a = r[0], b = r[1], c = r[2];
    const double w[6] = {a * b, a * b, b * c, b * c, c * a, c * a};
    double total = 0.0;
#pragma unroll
    for (int f = 0; f < 6; ++f) total += w[f];
    const double t = ((double)j + 0.5) / (double)n;
    int face = 0;
    double run = 0.0;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        run += w[f];
        if (t >= run / total) face = f + 1;
    }
    double ip;
    const double s1 = 2.0 * modf(0.5 + (double)j * 0.7548776662466927, &ip) - 1.0;
    const double s2 = 2.0 * modf(0.5 + (double)j * 0.5698402909980532, &ip) - 1.0;
    const double sg = (face & 1) ? -1.0 : 1.0;
    double v[3];
    if (face < 2) { v[0] = s1; v[1] = s2; v[2] = sg; }
    else if (face < 4) { v[0] = sg; v[1] = s1; v[2] = s2; }
    else { v[0] = s2; v[1] = sg; v[2] = s1; }
    const double side[3] = {a, b, c};
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] = (float)((v[i] * side[i]) / (side[i] + 1e-6));
}

// unit parameter of sample j of n on primitive `kind`: the point is V (r * u) + c for both kinds
__device__ __forceinline__ void unit_param(int kind, int j, int n, const float r[3], float u[3])
{
    if (kind == KIND_CUBOID) { cuboid_unit(j, n, r, u); return; }
    float cu, su, cv, sv;
    fib_dir(j, n, cu, su, cv, sv);
    u[0] = cu * sv; u[1] = su * sv; u[2] = cv;
}

constexpr int NN_TILE_T = 1024;

// Sample s of shape b: point on ellipsoid k(s), nearest target, squared distance (src/utils.py:413-416).
// Brute-force exact search (10^4 samples x 5 10^3 targets per shape): VALU-bound, so the inner loop is built for the
// vector pipe -- every thread carries TWO samples (s and s + 256) through the target tile, their distances are computed
// with packed fp32 instructions (one LDS read and 6 packed operations for two distances instead of 2 x 6 scalar ones;
// same operation order d = fma(dz, dz, fma(dy, dy, dx * dx)) and the same first-minimum rule as before, so the
// neighbours are the ones the scalar loop found), four targets per trip.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sample_point(int kind, const float *__restrict__ r, const float *__restrict__ V,
                                             const float *__restrict__ c, const int32_t *__restrict__ n_k,
                                             const int32_t *__restrict__ off, int KM, int b, int s, float &px, float &py,
                                             float &pz)
{
    int k = 0;
    while (k + 1 < KM && s >= off[k + 1]) ++k;
    const size_t sl = (size_t)b * KM + k;
    const float rk[3] = {r[sl * 3], r[sl * 3 + 1], r[sl * 3 + 2]};
    float ex, ey, ez;
    if (kind == KIND_CUBOID) {
        float u[3];
        cuboid_unit(s - off[k], n_k[sl], rk, u);
        ex = u[0] * rk[0]; ey = u[1] * rk[1]; ez = u[2] * rk[2];
    } else {
        float cu, su, cv, sv;
        fib_dir(s - off[k], n_k[sl], cu, su, cv, sv);
        ex = rk[0] * cu * sv; ey = rk[1] * su * sv; ez = rk[2] * cv;
    }
    const float *Vk = V + sl * 9;
    px = Vk[0] * ex + Vk[1] * ey + Vk[2] * ez + c[sl * 3];
    py = Vk[3] * ex + Vk[4] * ey + Vk[5] * ez + c[sl * 3 + 1];
    pz = Vk[6] * ex + Vk[7] * ey + Vk[8] * ez + c[sl * 3 + 2];
}

// The targets are cut into NN_SPLIT ranges (blockIdx.z): 4 x as many workgroups (624 -> 2496 at B = 24: the search is a
// long serial loop per thread, and 2.4 workgroups per CU left the chip half idle at the end); every range writes its
// (distance, index) candidate and sample_nn_pick_kernel keeps the first minimum over the ranges in ascending order --
// the same neighbour the unsplit loop finds.
constexpr int NN_SPLIT = 8;

__global__ __launch_bounds__(256) void sample_nn_fwd_kernel(
    int kind, const float *__restrict__ r, const float *__restrict__ V, const float *__restrict__ c,
    const int32_t *__restrict__ n_k, const int32_t *__restrict__ off_k, int KM, const float *__restrict__ tgt,
    int M, int cap, float *__restrict__ ws)
{
    __shared__ float4 s_t[NN_TILE_T];
    const int b = blockIdx.y;
    const int s0 = blockIdx.x * 512 + threadIdx.x, s1 = s0 + 256;
    const int32_t *off = off_k + (size_t)b * (KM + 1);
    const int total = off[KM];
    if (blockIdx.x * 512 >= total) return;  // block-uniform
    const bool act0 = s0 < total, act1 = s1 < total;
    float p0x = 0.f, p0y = 0.f, p0z = 0.f, p1x = 0.f, p1y = 0.f, p1z = 0.f;
    if (act0) sample_point(kind, r, V, c, n_k, off, KM, b, s0, p0x, p0y, p0z);
    if (act1) sample_point(kind, r, V, c, n_k, off, KM, b, s1, p1x, p1y, p1z);
    float best0 = INFINITY, best1 = INFINITY;
    int bi0 = 0, bi1 = 0;
    const float *T = tgt + (size_t)b * M * 3;
    const int chunk = (M + NN_SPLIT - 1) / NN_SPLIT;
    const int m_lo = blockIdx.z * chunk, m_hi = min(M, m_lo + chunk);
    for (int base = m_lo; base < m_hi; base += NN_TILE_T) {
        const int tn = min(NN_TILE_T, m_hi - base);
        __syncthreads();
        for (int i = threadIdx.x; i < NN_TILE_T; i += 256) {
            // rows beyond the tile repeat its last target: a duplicate never wins the strict comparison
            const int ii = base + (i < tn ? i : tn - 1);
            s_t[i] = make_float4(T[(size_t)ii * 3], T[(size_t)ii * 3 + 1], T[(size_t)ii * 3 + 2], 0.f);
        }
        __syncthreads();
        const int tn4 = (tn + 3) & ~3;
        for (int i = 0; i < tn4; i += 4) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = s_t[i + u];
            float d0[4], d1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (scalar fp32 on purpose: the packed v_pk_*_f32 forms of the same six operations measured slower here)
                const float ax = p0x - t[u].x, ay = p0y - t[u].y, az = p0z - t[u].z;
                const float bx = p1x - t[u].x, by = p1y - t[u].y, bz = p1z - t[u].z;
                d0[u] = fmaf(az, az, fmaf(ay, ay, ax * ax));
                d1[u] = fmaf(bz, bz, fmaf(by, by, bx * bx));
            }
            // one comparison per group of four targets: the running best keeps (distance, first target of the group); which of
            // the four it was is found once, after the loop (three vector instructions per group instead of twelve)
            const float m0 = fminf(fminf(d0[0], d0[1]), fminf(d0[2], d0[3]));
            const float m1 = fminf(fminf(d1[0], d1[1]), fminf(d1[2], d1[3]));
            if (m0 < best0) { best0 = m0; bi0 = base + i; }
            if (m1 < best1) { best1 = m1; bi1 = base + i; }
        }
    }
    // the winner inside its group: the same six operations on the same operands give the same bits; the first of the four that
    // equals the group's minimum is the first minimum of the range (strict comparisons between groups keep the earliest group).
    // A group that straddles the end of the range repeats the last target there: a repeat never comes before the original.
    {
        auto pick = [&](float px, float py, float pz, float best, int g0) {
            int w = g0;
#pragma unroll
            for (int u = 3; u >= 0; --u) {
                const int ii = min(g0 + u, m_hi - 1);
                const float ax = px - T[(size_t)ii * 3], ay = py - T[(size_t)ii * 3 + 1], az = pz - T[(size_t)ii * 3 + 2];
                if (fmaf(az, az, fmaf(ay, ay, ax * ax)) == best) w = ii;
            }
            return w;
        };
        if (act0 && best0 < INFINITY) bi0 = pick(p0x, p0y, p0z, best0, bi0);
        if (act1 && best1 < INFINITY) bi1 = pick(p1x, p1y, p1z, best1, bi1);
    }
    // (a padded duplicate can only tie with the real last target of the range, which came first: never taken)
    float2 *cand = reinterpret_cast<float2 *>(ws) + ((size_t)b * NN_SPLIT + blockIdx.z) * cap;
    if (act0) cand[s0] = make_float2(best0, __int_as_float(bi0));
    if (act1) cand[s1] = make_float2(best1, __int_as_float(bi1));
}

// first minimum over the target ranges (ascending), nearest index out, sum of the squared distances per shape
__global__ __launch_bounds__(256) void sample_nn_pick_kernel(const float *__restrict__ ws, const int32_t *__restrict__ off_k,
                                                             int KM, int M, int cap, int32_t *__restrict__ nn_idx,
                                                             float *__restrict__ sum_o)
{
    __shared__ float s_red[4];
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    const int total = off_k[(size_t)b * (KM + 1) + KM];
    if (blockIdx.x * 256 >= total) return;  // block-uniform
    const bool act = s < total;
    float best = INFINITY;
    int bi = 0;
    if (act) {
        const int chunk = (M + NN_SPLIT - 1) / NN_SPLIT;
#pragma unroll
        for (int z = 0; z < NN_SPLIT; ++z) {
            if (z * chunk >= M) break;
            const float2 e = reinterpret_cast<const float2 *>(ws)[((size_t)b * NN_SPLIT + z) * cap + s];
            if (e.x < best) { best = e.x; bi = __float_as_int(e.y); }
        }
        nn_idx[(size_t)b * cap + s] = min(bi, M - 1);
    }
    float v = act ? best : 0.f;
    v = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sum_o + b, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

__global__ __launch_bounds__(256) void sample_nn_bwd_kernel(
    int kind, const float *__restrict__ r, const float *__restrict__ V, const float *__restrict__ c,
    const int32_t *__restrict__ n_k, const int32_t *__restrict__ off_k, int KM, const float *__restrict__ tgt,
    int M, int cap, const int32_t *__restrict__ nn_idx, const float *__restrict__ gscale,
    float *__restrict__ g_r, float *__restrict__ g_V, float *__restrict__ g_c)
{
    __shared__ float s_acc[KM_MAX * 15];
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int32_t *off = off_k + (size_t)b * (KM + 1);
    const int total = off[KM];
    if (blockIdx.x * 256 >= total) return;
    for (int i = threadIdx.x; i < KM * 15; i += 256) s_acc[i] = 0.f;
    __syncthreads();
    if (s < total) {
        int k = 0;
        while (k + 1 < KM && s >= off[k + 1]) ++k;
        const size_t sl = (size_t)b * KM + k;
        const float rk[3] = {r[sl * 3], r[sl * 3 + 1], r[sl * 3 + 2]};
        float dir[3];
        unit_param(kind, s - off[k], n_k[sl], rk, dir);
        const float e[3] = {rk[0] * dir[0], rk[1] * dir[1], rk[2] * dir[2]};
        const float *Vk = V + sl * 9;
        const float *t = tgt + ((size_t)b * M + nn_idx[(size_t)b * cap + s]) * 3;
        float gs[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float p = Vk[i * 3] * e[0] + Vk[i * 3 + 1] * e[1] + Vk[i * 3 + 2] * e[2] + c[sl * 3 + i];
            gs[i] = gscale[b] * 2.0f * (p - t[i]);
        }
        float *acc = s_acc + k * 15;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float ge = Vk[j] * gs[0] + Vk[3 + j] * gs[1] + Vk[6 + j] * gs[2];  // (V^T g_s)[j]
            atomicAdd(acc + j, ge * dir[j]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) atomicAdd(acc + 3 + i * 3 + j, gs[i] * e[j]);
            atomicAdd(acc + 12 + i, gs[i]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KM * 15; i += 256) {
        const float v = s_acc[i];
        if (v == 0.f) continue;
        const int k = i / 15, j = i % 15;
        const size_t sl = (size_t)b * KM + k;
        if (j < 3) unsafeAtomicAdd(g_r + sl * 3 + j, v);
        else if (j < 12) unsafeAtomicAdd(g_V + sl * 9 + (j - 3), v);
        else unsafeAtomicAdd(g_c + sl * 3 + (j - 12), v);
    }
}

extern "C" {

int prifit_fit_state_floats(void) { return FIT_STATE; }

int prifit_ellipsoid_fit_fwd(const float *points, const float *W, const int32_t *count, const float *rnd,
                             long long rnd_stride_b, long long rnd_stride_k, int canonical_signs, int B, int N,
                             int KM, float *r, float *V, float *c, int32_t *valid, float *state, void *stream)
{
    if (!points || !W || !count || !rnd || !r || !V || !c || !valid || !state || B <= 0 || N <= 0 || KM <= 0 ||
        KM > KM_MAX)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(ellipsoid_fit_fwd_kernel, dim3(KM, B), dim3(256), 0, as_stream(stream), points, W, count,
                       rnd, rnd_stride_b, rnd_stride_k, canonical_signs, N, KM, r, V, c, valid, state);
    return prifit_check_launch();
}

int prifit_ellipsoid_fit_bwd(const float *points, const float *W, const int32_t *count, const int32_t *valid,
                             const float *rnd, long long rnd_stride_b, long long rnd_stride_k, const float *state,
                             const float *g_r, const float *g_V, const float *g_c, int B, int N, int KM, float *gW,
                             void *stream)
{
    if (!points || !W || !count || !valid || !rnd || !state || !g_r || !g_V || !g_c || !gW || B <= 0 || N <= 0 ||
        KM <= 0 || KM > KM_MAX)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(ellipsoid_fit_bwd_kernel, dim3(KM, B), dim3(256), 0, as_stream(stream), points, W, count,
                       valid, rnd, rnd_stride_b, rnd_stride_k, state, g_r, g_V, g_c, N, KM, gW);
    return prifit_check_launch();
}

static int impl_ellipsoid_sdf_fwd(int kind, const float *targets, int B, int M, const float *r, const float *V, const float *c,
                             const int32_t *valid, int KM, int32_t *arg, float *fval, float *sum_sq, void *stream)
{
    if (!targets || !r || !V || !c || !valid || !arg || !fval || !sum_sq || B <= 0 || M <= 0 || KM <= 0 ||
        KM > KM_MAX)
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sum_sq, 0, sizeof(float) * B, st) != hipSuccess) return PRIFIT_ELAUNCH;
    hipLaunchKernelGGL(sdf_fwd_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, kind, targets, M, r, V, c, valid, KM,
                       arg, fval, sum_sq);
    return prifit_check_launch();
}

static int impl_ellipsoid_sdf_bwd(int kind, const float *targets, int B, int M, const float *r, const float *V, const float *c,
                             const int32_t *arg, const float *gscale, int KM, float *g_r, float *g_V, float *g_c,
                             void *stream)
{
    if (!targets || !r || !V || !c || !arg || !gscale || !g_r || !g_V || !g_c || B <= 0 || M <= 0 || KM <= 0 ||
        KM > KM_MAX)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(sdf_bwd_kernel, dim3((M + 255) / 256, B), dim3(256), 0, as_stream(stream), kind, targets, M, r, V,
                       c, arg, gscale, KM, g_r, g_V, g_c);
    return prifit_check_launch();
}

static int impl_ellipsoid_sdf_matrix_fwd(int kind, const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, int KM, float *sdf, void *stream)
{
    if (!points || !r || !V || !c || !valid || !sdf || B <= 0 || M <= 0 || KM <= 0 || KM > KM_MAX)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(sdf_matrix_fwd_kernel, dim3((M + 255) / 256, B), dim3(256), 0, as_stream(stream), kind, points, M,
                       r, V, c, valid, KM, sdf);
    return prifit_check_launch();
}

static int impl_ellipsoid_sdf_matrix_bwd(int kind, const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, const float *g_sdf, int KM, float *g_r,
                                    float *g_V, float *g_c, void *stream)
{
    if (!points || !r || !V || !c || !valid || !g_sdf || !g_r || !g_V || !g_c || B <= 0 || M <= 0 || KM <= 0 ||
        KM > KM_MAX)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(sdf_matrix_bwd_kernel, dim3((M + 255) / 256, B), dim3(256), 0, as_stream(stream), kind, points, M,
                       r, V, c, valid, g_sdf, KM, g_r, g_V, g_c);
    return prifit_check_launch();
}

static int impl_sample_budget(int kind, const float *r, const int32_t *valid, int B, int KM, int cap, int32_t *n, int32_t *off,
                         void *stream)
{
    if (!r || !valid || !n || !off || B <= 0 || KM <= 0 || KM > KM_MAX || cap <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(sample_budget_kernel, dim3(B), dim3(64), 0, as_stream(stream), kind, r, valid, KM, cap, n, off);
    return prifit_check_launch();
}

static int impl_sample_nn_fwd(int kind, const float *r, const float *V, const float *c, const int32_t *n, const int32_t *off,
                         int B, int KM, const float *targets, int M, int cap, int32_t *nn_idx, float *sum_d2,
                         float *workspace, void *stream)
{
    if (!r || !V || !c || !n || !off || !targets || !nn_idx || !sum_d2 || !workspace || B <= 0 || KM <= 0 || KM > KM_MAX ||
        M <= 0 || cap <= 0 || ((uintptr_t)workspace & 7))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sum_d2, 0, sizeof(float) * B, st) != hipSuccess) return PRIFIT_ELAUNCH;
    hipLaunchKernelGGL(sample_nn_fwd_kernel, dim3((cap + 511) / 512, B, NN_SPLIT), dim3(256), 0, st, kind, r, V, c, n, off,
                       KM, targets, M, cap, workspace);
    hipLaunchKernelGGL(sample_nn_pick_kernel, dim3((cap + 255) / 256, B), dim3(256), 0, st, workspace, off, KM, M, cap,
                       nn_idx, sum_d2);
    return prifit_check_launch();
}

static int impl_sample_nn_bwd(int kind, const float *r, const float *V, const float *c, const int32_t *n, const int32_t *off,
                         int B, int KM, const float *targets, int M, int cap, const int32_t *nn_idx,
                         const float *gscale, float *g_r, float *g_V, float *g_c, void *stream)
{
    if (!r || !V || !c || !n || !off || !targets || !nn_idx || !gscale || !g_r || !g_V || !g_c || B <= 0 ||
        KM <= 0 || KM > KM_MAX || M <= 0 || cap <= 0)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(sample_nn_bwd_kernel, dim3((cap + 255) / 256, B), dim3(256), 0, as_stream(stream), kind, r, V, c,
                       n, off, KM, targets, M, cap, nn_idx, gscale, g_r, g_V, g_c);
    return prifit_check_launch();
}

int prifit_ellipsoid_sdf_fwd(const float *targets, int B, int M, const float *r, const float *V, const float *c,
                             const int32_t *valid, int KM, int32_t *arg, float *fval, float *sum_sq, void *stream)
{
    return impl_ellipsoid_sdf_fwd(KIND_ELLIPSOID, targets, B, M, r, V, c, valid, KM, arg, fval, sum_sq, stream);
}

int prifit_cuboid_sdf_fwd(const float *targets, int B, int M, const float *r, const float *V, const float *c,
                             const int32_t *valid, int KM, int32_t *arg, float *fval, float *sum_sq, void *stream)
{
    return impl_ellipsoid_sdf_fwd(KIND_CUBOID, targets, B, M, r, V, c, valid, KM, arg, fval, sum_sq, stream);
}

int prifit_ellipsoid_sdf_bwd(const float *targets, int B, int M, const float *r, const float *V, const float *c,
                             const int32_t *arg, const float *gscale, int KM, float *g_r, float *g_V, float *g_c,
                             void *stream)
{
    return impl_ellipsoid_sdf_bwd(KIND_ELLIPSOID, targets, B, M, r, V, c, arg, gscale, KM, g_r, g_V, g_c, stream);
}

int prifit_cuboid_sdf_bwd(const float *targets, int B, int M, const float *r, const float *V, const float *c,
                             const int32_t *arg, const float *gscale, int KM, float *g_r, float *g_V, float *g_c,
                             void *stream)
{
    return impl_ellipsoid_sdf_bwd(KIND_CUBOID, targets, B, M, r, V, c, arg, gscale, KM, g_r, g_V, g_c, stream);
}

int prifit_ellipsoid_sdf_matrix_fwd(const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, int KM, float *sdf, void *stream)
{
    return impl_ellipsoid_sdf_matrix_fwd(KIND_ELLIPSOID, points, B, M, r, V, c, valid, KM, sdf, stream);
}

int prifit_cuboid_sdf_matrix_fwd(const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, int KM, float *sdf, void *stream)
{
    return impl_ellipsoid_sdf_matrix_fwd(KIND_CUBOID, points, B, M, r, V, c, valid, KM, sdf, stream);
}

int prifit_ellipsoid_sdf_matrix_bwd(const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, const float *g_sdf, int KM, float *g_r,
                                    float *g_V, float *g_c, void *stream)
{
    return impl_ellipsoid_sdf_matrix_bwd(KIND_ELLIPSOID, points, B, M, r, V, c, valid, g_sdf, KM, g_r, g_V, g_c, stream);
}

int prifit_cuboid_sdf_matrix_bwd(const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, const float *g_sdf, int KM, float *g_r,
                                    float *g_V, float *g_c, void *stream)
{
    return impl_ellipsoid_sdf_matrix_bwd(KIND_CUBOID, points, B, M, r, V, c, valid, g_sdf, KM, g_r, g_V, g_c, stream);
}

int prifit_sample_budget(const float *r, const int32_t *valid, int B, int KM, int cap, int32_t *n, int32_t *off,
                         void *stream)
{
    return impl_sample_budget(KIND_ELLIPSOID, r, valid, B, KM, cap, n, off, stream);
}

int prifit_cuboid_sample_budget(const float *r, const int32_t *valid, int B, int KM, int cap, int32_t *n, int32_t *off,
                         void *stream)
{
    return impl_sample_budget(KIND_CUBOID, r, valid, B, KM, cap, n, off, stream);
}

long long prifit_sample_nn_workspace_floats(int B, int cap) { return 2LL * NN_SPLIT * B * cap; }

int prifit_sample_nn_fwd(const float *r, const float *V, const float *c, const int32_t *n, const int32_t *off,
                         int B, int KM, const float *targets, int M, int cap, int32_t *nn_idx, float *sum_d2,
                         float *workspace, void *stream)
{
    return impl_sample_nn_fwd(KIND_ELLIPSOID, r, V, c, n, off, B, KM, targets, M, cap, nn_idx, sum_d2, workspace, stream);
}

int prifit_cuboid_sample_nn_fwd(const float *r, const float *V, const float *c, const int32_t *n, const int32_t *off,
                         int B, int KM, const float *targets, int M, int cap, int32_t *nn_idx, float *sum_d2,
                         float *workspace, void *stream)
{
    return impl_sample_nn_fwd(KIND_CUBOID, r, V, c, n, off, B, KM, targets, M, cap, nn_idx, sum_d2, workspace, stream);
}

int prifit_sample_nn_bwd(const float *r, const float *V, const float *c, const int32_t *n, const int32_t *off,
                         int B, int KM, const float *targets, int M, int cap, const int32_t *nn_idx,
                         const float *gscale, float *g_r, float *g_V, float *g_c, void *stream)
{
    return impl_sample_nn_bwd(KIND_ELLIPSOID, r, V, c, n, off, B, KM, targets, M, cap, nn_idx, gscale, g_r, g_V, g_c, stream);
}

int prifit_cuboid_sample_nn_bwd(const float *r, const float *V, const float *c, const int32_t *n, const int32_t *off,
                         int B, int KM, const float *targets, int M, int cap, const int32_t *nn_idx,
                         const float *gscale, float *g_r, float *g_V, float *g_c, void *stream)
{
    return impl_sample_nn_bwd(KIND_CUBOID, r, V, c, n, off, B, KM, targets, M, cap, nn_idx, gscale, g_r, g_V, g_c, stream);
}

}  // extern "C"
