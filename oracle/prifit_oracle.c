/*
 * prifit_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Scalar C restatement of the index-producing ops of the reference's PointNet++ stack, written
 * so that every floating-point operation and its rounding is explicit.  The integer results
 * (sampled / grouped / neighbour indices) must be BIT-EXACT between this file, the reference's
 * PyTorch-CPU path and the HIP kernels in prifit_amd/csrc.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Reference sites restated here (paths relative to the upstream repo):
 *   models/pointnet_util.py:19-40   square_distance   (expanded form, fp32, matmul with K=3)
 *   models/pointnet_util.py:63-84   farthest_point_sample
 *   models/pointnet_util.py:87-107  query_ball_point
 *   models/pointnet_util.py:291-293 3-NN selection inside PointNetFeaturePropagation.forward
 *   src/dgcnn.py:9-27               knn (top-k of -||xi-xj||^2)
 *
 * Parity pinned: yes -- tests/golden/ npz files hold outputs captured from the reference itself
 * (oracle/make_golden.py) and tests/test_oracle_golden.py checks this file against them.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* (x*x + y*y) + z*z with one rounding per operation: torch.sum(t ** 2, -1) on a [.., 3] tensor. */
static inline float norm2_3(const float *p)
{
    float a = p[0] * p[0];
    float b = p[1] * p[1];
    float c = p[2] * p[2];
    float ab = a + b;
    return ab + c;
}

/* The K=3 sgemm dot product as the CPU BLAS evaluates it: an fma chain seeded by a product. */
static inline float dot3(const float *q, const float *p)
{
    float t = q[0] * p[0];
    t = fmaf(q[1], p[1], t);
    t = fmaf(q[2], p[2], t);
    return t;
}

/* pointnet_util.py:37-39: dist = -2*matmul; dist += |src|^2; dist += |dst|^2 */
static inline float sqdist_expanded(const float *src, float ss, const float *dst, float dd)
{
    float m2 = -2.0f * dot3(src, dst); /* exact scaling */
    float t = m2 + ss;
    return t + dd;
}

void orc_square_distance(const float *src, const float *dst, int B, int S, int N, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            const float *q = src + ((size_t)b * S + s) * 3;
            float ss = norm2_3(q);
            for (int n = 0; n < N; ++n) {
                const float *p = dst + ((size_t)b * N + n) * 3;
                out[((size_t)b * S + s) * N + n] = sqdist_expanded(q, ss, p, norm2_3(p));
            }
        }
}

/* pointnet_util.py:63-84.  start[b] replaces torch.randint (line 75).  Direct-form distance
 * ((dx*dx + dy*dy) + dz*dz), running minimum initialised to 1e10, first maximum wins. */
void orc_fps(const float *xyz, int B, int N, int npoint, const int64_t *start, int64_t *out)
{
    float *mind = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *P = xyz + (size_t)b * N * 3;
        for (int n = 0; n < N; ++n) mind[n] = 1e10f;
        int64_t far = start[b];
        for (int i = 0; i < npoint; ++i) {
            out[(size_t)b * npoint + i] = far;
            const float cx = P[far * 3 + 0], cy = P[far * 3 + 1], cz = P[far * 3 + 2];
            float best = -INFINITY;
            int64_t besti = 0;
            for (int n = 0; n < N; ++n) {
                float dx = P[n * 3 + 0] - cx;
                float dy = P[n * 3 + 1] - cy;
                float dz = P[n * 3 + 2] - cz;
                float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                float d = (xx + yy) + zz;
                if (d < mind[n]) mind[n] = d;
                if (mind[n] > best) { best = mind[n]; besti = n; }
            }
            far = besti;
        }
    }
    free(mind);
}

/* pointnet_util.py:87-107.  The reference marks out-of-ball points with N, sorts, keeps the first
 * nsample and overwrites the N's with the first entry: i.e. the first `nsample` indices n (ascending)
 * with NOT(d > r2), padded with the first of them.  A query with no in-ball point keeps N everywhere
 * (group_first == N), which we reproduce. */
void orc_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample,
                    int64_t *out)
{
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            const float *q = new_xyz + ((size_t)b * S + s) * 3;
            float qq = norm2_3(q);
            int64_t *o = out + ((size_t)b * S + s) * nsample;
            int cnt = 0;
            for (int n = 0; n < N && cnt < nsample; ++n) {
                const float *p = xyz + ((size_t)b * N + n) * 3;
                float d = sqdist_expanded(q, qq, p, norm2_3(p));
                if (!(d > r2)) o[cnt++] = n;
            }
            int64_t first = cnt ? o[0] : (int64_t)N;
            for (int k = cnt; k < nsample; ++k) o[k] = first;
        }
}

/* pointnet_util.py:291-293: dists = square_distance(xyz1, xyz2); sort ascending; first three.
 * Ties resolve to the lower index (stable order). */
void orc_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx, float *dist)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const float *q = xyz1 + ((size_t)b * N + n) * 3;
            float qq = norm2_3(q);
            float bd[3] = {INFINITY, INFINITY, INFINITY};
            int64_t bi[3] = {0, 0, 0};
            for (int s = 0; s < S; ++s) {
                const float *p = xyz2 + ((size_t)b * S + s) * 3;
                float d = sqdist_expanded(q, qq, p, norm2_3(p));
                if (d < bd[2]) {
                    int k = 2;
                    while (k > 0 && d < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                    bd[k] = d; bi[k] = s;
                }
            }
            for (int k = 0; k < 3; ++k) {
                idx[((size_t)b * N + n) * 3 + k] = bi[k];
                dist[((size_t)b * N + n) * 3 + k] = bd[k];
            }
        }
}

/* src/dgcnn.py:9-27: inner = -2 x^T x; xx = sum(x**2); pairwise = -xx - inner - xx^T;
 * idx = topk(k) (largest).  x is [B, C, N]; here points are passed channels-last [B, N, C].
 * Evaluated as ((-xx_i) - inner_ij) - xx_j with inner = -2 * (fma chain over c).  Ties -> lower index. */
void orc_knn(const float *x, int B, int N, int C, int k, int64_t *idx)
{
    float *xx = (float *)malloc(sizeof(float) * (size_t)N);
    float *val = (float *)malloc(sizeof(float) * (size_t)k);
    for (int b = 0; b < B; ++b) {
        const float *X = x + (size_t)b * N * C;
        for (int n = 0; n < N; ++n) {
            float s = X[(size_t)n * C] * X[(size_t)n * C];
            for (int c = 1; c < C; ++c) s += X[(size_t)n * C + c] * X[(size_t)n * C + c];
            xx[n] = s;
        }
        for (int i = 0; i < N; ++i) {
            int64_t *o = idx + ((size_t)b * N + i) * k;
            int cnt = 0;
            for (int j = 0; j < N; ++j) {
                float t = X[(size_t)i * C] * X[(size_t)j * C];
                for (int c = 1; c < C; ++c) t = fmaf(X[(size_t)i * C + c], X[(size_t)j * C + c], t);
                float inner = -2.0f * t;
                float v = (-xx[i] - inner) - xx[j];
                if (cnt < k || v > val[cnt - 1]) {
                    int p = cnt < k ? cnt++ : k - 1;
                    while (p > 0 && v > val[p - 1]) { val[p] = val[p - 1]; o[p] = o[p - 1]; --p; }
                    val[p] = v; o[p] = j;
                }
            }
        }
    }
    free(xx);
    free(val);
}
