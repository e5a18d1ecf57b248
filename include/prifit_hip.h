/*
 * prifit_hip.h -- C ABI of libprifit_hip.so, the MI355X (gfx950) backend of the PRIFIT hot path.
 *
 * The upstream reference (Hippogriff/prifit) is pure Python on PyTorch: it has no FFI / plugin
 * interface, its "kernels" are ATen ops composed in Python.  The drop-in boundary is therefore
 * the Python call surface (models/pointnet_util.py, models/pointnet2_part_seg_msg.py,
 * convex_loss.py, src/mean_shift.py, src/ellipsoid_fitting.py); this header is the C ABI that
 * the build's mirror of that surface (package `prifit_amd`) binds with ctypes.  Each entry
 * point cites the reference site (file:line, relative to the upstream repo) it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (PRIFIT_E*); nothing throws across the ABI;
 *   - all pointers are caller-owned DEVICE pointers (hipMalloc'ed; PyTorch allocates them),
 *     contiguous row-major, fp32 / int32 / int64 as stated; the library never allocates, frees,
 *     retains or synchronises; scratch is passed in by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are only
 *     enqueued, so every entry point is legal inside a HIP graph capture;
 *   - the library holds no mutable global state and is re-entrant;
 *   - layouts are "channels-last": clouds [B, N, 3], feature tables [B, N, C], activation
 *     matrices [P, ld] with P = number of positions (B*S*K grouped samples or B*N points).
 */
#ifndef PRIFIT_HIP_H
#define PRIFIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRIFIT_OK 0
#define PRIFIT_EINVAL (-1)   /* bad argument (shape / alignment / unsupported size) */
#define PRIFIT_ELAUNCH (-2)  /* hipGetLastError() != hipSuccess after the launch   */

/* Library identification: returns 10000*major + 100*minor + patch; *arch (may be NULL) receives a
 * static string naming the code object target ("gfx950"). */
int prifit_version(const char **arch);

/* ------------------------------------------------------------------------------------------ */
/* PointNet++ index ops (bit-exact with the reference's PyTorch-CPU results)                    */
/* ------------------------------------------------------------------------------------------ */

/* Farthest point sampling.  Replaces models/pointnet_util.py:63-84 (farthest_point_sample).
 * xyz [B,N,3]; start_idx [B] replaces the torch.randint of line 75; out_idx [B,npoint] int64;
 * new_xyz [B,npoint,3] (may be NULL) receives the gathered centroids (index_points, :43-60).
 * One workgroup per shape; N <= 4096. */
int prifit_fps(const float *xyz, int B, int N, int npoint, const int64_t *start_idx,
               int64_t *out_idx, float *new_xyz, void *stream);

/* Multi-radius ball query.  Replaces models/pointnet_util.py:87-107 (query_ball_point, with
 * square_distance :19-40 fused) for R <= 4 radii in one pass over the points.
 * xyz [B,N,3], new_xyz [B,S,3]; radius2[r] = (float)(radius*radius) computed by the caller in
 * double then rounded to fp32 (that is what the reference's comparison does); nsample[r] <= 1024;
 * out[r] is a device pointer to [B,S,nsample[r]] of int32 (idx64 == 0) or int64 (idx64 != 0).
 * radius2 / nsample / out are HOST arrays of length R read before the call returns. */
int prifit_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, int R,
                      const float *radius2, const int *nsample, void *const *out, int idx64,
                      void *stream);

/* Three nearest neighbours + inverse-distance weights.  Replaces the selection and weighting
 * inside PointNetFeaturePropagation.forward, models/pointnet_util.py:291-297.
 * xyz1 [B,N,3] (queries), xyz2 [B,S,3], S >= 3; idx [B,N,3] int32 (ascending distance, ties to the
 * lower index), dist [B,N,3] expanded-form squared distances (may be NULL), weight [B,N,3]. */
int prifit_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int32_t *idx,
                    float *dist, float *weight, void *stream);

/* Expanded-form squared distances, models/pointnet_util.py:19-40 (bitwise: -2*dot + |s|^2 + |d|^2).
 * src [B,S,3], dst [B,N,3] -> out [B,S,N]. */
int prifit_square_distance(const float *src, const float *dst, int B, int S, int N, float *out,
                           void *stream);

/* ------------------------------------------------------------------------------------------ */
/* grouping / interpolation (bandwidth kernels)                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Gather grouped rows.  Replaces index_points + centroid subtraction + cat of
 * models/pointnet_util.py:243-249 (MSG: [features, rel_xyz]) and :127-133 (SSG: [rel_xyz, features]).
 * feat [B,N,C] (NULL when C == 0), xyz [B,N,3], new_xyz [B,S,3], idx [B,S,K] int32 (entries >= N
 * produce a zero row), out [B*S*K, ld_out] with ld_out >= C+3; columns >= C+3 are zero-filled.
 * order 0: out = [feat(C), rel(3), 0...]; order 1: out = [rel(3), feat(C), 0...]. */
int prifit_group_gather(const float *feat, const float *xyz, const float *new_xyz,
                        const int32_t *idx, int B, int N, int S, int K, int C, int order,
                        int ld_out, float *out, void *stream);

/* Backward of the feature part of prifit_group_gather (autograd of index_points, :59):
 * dfeat[b, idx[b,s,k], c] += gout[(b,s,k), col0 + c].  dfeat [B,N,C] must be initialised by the
 * caller (zeros, or a gradient to accumulate into). */
int prifit_group_scatter_add(const float *gout, int ld_gout, int col0, const int32_t *idx, int B,
                             int N, int S, int K, int C, float *dfeat, void *stream);

/* out[(b,n), col0 + c] = sum_j weight[b,n,j] * points2[b, idx[b,n,j], c]
 * (models/pointnet_util.py:298).  points2 [B,S,C], out rows have stride ld_out. */
int prifit_three_interpolate(const float *points2, const int32_t *idx, const float *weight, int B,
                             int N, int S, int C, int ld_out, int col0, float *out, void *stream);

/* Backward of prifit_three_interpolate w.r.t. points2 (dpoints2 [B,S,C] initialised by the caller). */
int prifit_three_interpolate_bwd(const float *gout, int ld_gout, int col0, const int32_t *idx,
                                 const float *weight, int B, int N, int S, int C, float *dpoints2,
                                 void *stream);

/* ------------------------------------------------------------------------------------------ */
/* dense contraction on the matrix cores (fp32 in / fp32 accumulate MFMA, exact f32)            */
/* ------------------------------------------------------------------------------------------ */

#define PRIFIT_GEMM_NT 0 /* C[M,N] = A[M,K] . B[N,K]^T  (both operands k-contiguous)            */
#define PRIFIT_GEMM_NN 1 /* C[M,N] = A[M,K] . B[K,N]                                             */
#define PRIFIT_GEMM_TN 2 /* C[M,N] = A[K,M]^T . B[K,N]  (reduction over the leading index)      */
#define PRIFIT_EPI_NONE 0     /* C = acc (+ bias)                                                */
#define PRIFIT_EPI_CHORD 1    /* C = 2 - 2*acc            (src/mean_shift.py:154,168,185)        */
#define PRIFIT_EPI_MSKERNEL 2 /* C = exp(clamp(-(2-2*acc)/b^2/2, -13, 75)), b = epi_batch_scalar[z]
                                 (src/mean_shift.py:65-68 with src/guard.py:6-11)                 */

/* Batched GEMM with fused prologue/epilogue.  Replaces every conv1x1 of the shared per-position
 * MLPs (models/pointnet_util.py:195-199, :252-256, :310-313; models/pointnet2_part_seg_msg.py:88,
 * :109,:128) and their autograd, and the N x N x D products of src/mean_shift.py:65,73,154,168,185.
 *   lda/ldb/ldc: row strides; strideA/B/C: batch strides (elements); grid z = batch * splitk.
 *   a_scale/a_shift (both or neither): operand A is read as max(a*scale[c]+shift[c], 0) with c the
 *     index along A's contiguous dimension (k for NT/NN, m for TN) -- the train-mode BatchNorm+ReLU
 *     of the producing layer applied on load.  b_scale/b_shift: the same for B (c = k for NT, n else).
 *   bias [N] or NULL.  col_stats [ceil(M/tile_m)][2][N] or NULL: per-M-tile partial column sums
 *     and sums of squares of the stored C (batch == 1, splitk == 1 only).
 *   splitk > 1: the K range is split over workgroups and C is accumulated with float atomics
 *     (C must be initialised by the caller; epilogue must be PRIFIT_EPI_NONE).
 * 16-byte vector loads are used when pointers, strides and contiguous extents are multiples of 4
 * floats; any other shape takes a scalar-load path. */
int prifit_gemm_f32(int layout, int M, int N, int K, const float *A, long long lda, long long strideA,
                    const float *B, long long ldb, long long strideB, float *C, long long ldc,
                    long long strideC, int batch, int splitk, const float *a_scale,
                    const float *a_shift, const float *b_scale, const float *b_shift,
                    const float *bias, float *col_stats, int epilogue,
                    const float *epi_batch_scalar, void *stream);

/* Rows of C covered by one col_stats slab of prifit_gemm_f32 (its M tile). */
int prifit_gemm_tile_m(int N);

/* ------------------------------------------------------------------------------------------ */
/* train-mode BatchNorm + ReLU + group max-pool around the GEMMs                                */
/* (matrices are [P, ld] channels-last, C % 4 == 0, rows 16-byte aligned)                       */
/* ------------------------------------------------------------------------------------------ */

/* Rows reduced into one partial slab by the *_reduce / col_stats kernels below. */
int prifit_reduce_rows_per_slab(void);

/* slab [ceil(P/rows_per_slab)][2][C] = per-block column (sum, sum of squares) of Y. */
int prifit_col_stats(const float *Y, long long ld, int P, int C, float *slab, void *stream);

/* Batch statistics -> affine form of BatchNorm (torch.nn.BatchNorm{1,2}d in train mode, as used at
 * models/pointnet_util.py:198,254,312): mean/var over `count` positions from the partial slabs,
 * scale = gamma*invstd, shift = beta - mean*scale; running stats (may be NULL) updated with
 * `momentum` and the unbiased variance. */
int prifit_bn_finalize(const float *slab, int nslab, int C, double count, const float *gamma,
                       const float *beta, float eps, float momentum, float *running_mean,
                       float *running_var, float *scale, float *shift, float *mean, float *invstd,
                       void *stream);

/* out = max(Y*scale + shift, 0): F.relu(bn(.)) materialised (module outputs). */
int prifit_affine_relu(const float *Y, long long ldy, const float *scale, const float *shift, int P,
                       int C, float *out, long long ldo, void *stream);

/* torch.max(relu(bn(Y)), dim=K)[0] (models/pointnet_util.py:199,256): Y [G*K, ldy] -> out [G, ldo],
 * arg [G, C] = index k of the first maximum. */
int prifit_pool_fwd(const float *Y, long long ldy, const float *scale, const float *shift, int G, int K,
                    int C, float *out, long long ldo, int32_t *arg, void *stream);

/* Backward of relu(bn(Y)) given G = dL/d(relu output): partial slabs of m1 = sum(G*mask) and
 * m2 = sum(G*mask*yhat). */
int prifit_bn_relu_bwd_reduce(const float *G, long long ldg, const float *Y, long long ldy,
                              const float *scale, const float *shift, const float *mean,
                              const float *invstd, int P, int C, float *slab, void *stream);

/* The same partials when the gradient gp [G, ldgp] arrives through the group max-pool. */
int prifit_pool_bwd_reduce(const float *gp, long long ldgp, const float *Y, long long ldy,
                           const int32_t *arg, const float *scale, const float *shift,
                           const float *mean, const float *invstd, int G, int K, int C, float *slab,
                           void *stream);

/* m1, m2 -> dgamma, dbeta and the per-channel coefficients of dY = a*(G*mask) + b*Y + d
 * (training != 0: batch-stat BatchNorm backward; training == 0: running-stat affine). */
int prifit_bn_bwd_finalize(const float *slab, int nslab, int C, double count, int training,
                           const float *scale, const float *mean, const float *invstd, float *dgamma,
                           float *dbeta, float *coef_a, float *coef_b, float *coef_d, void *stream);

int prifit_bn_relu_bwd_apply(const float *G, long long ldg, const float *Y, long long ldy,
                             const float *scale, const float *shift, const float *coef_a,
                             const float *coef_b, const float *coef_d, int P, int C, float *dY,
                             long long ldd, void *stream);

int prifit_pool_bwd_apply(const float *gp, long long ldgp, const float *Y, long long ldy,
                          const int32_t *arg, const float *scale, const float *shift,
                          const float *coef_a, const float *coef_b, const float *coef_d, int G, int K,
                          int C, float *dY, long long ldd, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PRIFIT_HIP_H */
