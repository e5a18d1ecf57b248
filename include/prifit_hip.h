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

#ifdef __cplusplus
}
#endif
#endif /* PRIFIT_HIP_H */
