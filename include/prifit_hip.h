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

/* BatchNorm tails (round 6).  A kernel that produces the per-column sums of a BatchNorm layer can also FINALIZE them: pass one of
 * these descriptors (host structs, read at launch) instead of a statistics-slab pointer and the launch leaves the layer's
 * coefficients in `out` -- the separate prifit_bn_finalize / prifit_bn_bwd_finalize launch (and the slab) disappear.
 *   acc     [prifit_bn_tail_replicas()][2][C] doubles, ZERO on entry (left zero);  ticket  one int32, ZERO on entry (left zero)
 * prifit_bn_fwd: train-mode nn.BatchNorm of models/pointnet_util.py:195-197 / :250-252 / :310-313 -- out [4][C] = scale
 *   (gamma * invstd), shift (beta - mean * scale), mean, invstd; running statistics updated with `momentum` (unbiased variance)
 *   when the pointers are given; count = rows behind the sums.
 * prifit_bn_bwd: its autograd -- out [5][C] = dgamma, dbeta and the coefficients (a, b, d) of dY = a Gm + b Y + d.
 * A NULL descriptor (or acc == NULL) keeps the slab form of the entry point. */
int prifit_bn_tail_replicas(void);

typedef struct prifit_bn_fwd {
    double *acc;
    int32_t *ticket;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    float *out;
    double count;
    float eps, momentum;
} prifit_bn_fwd;

typedef struct prifit_bn_bwd {
    double *acc;
    int32_t *ticket;
    const float *scale, *mean, *invstd;
    float *out;
    double count;
    int training;
} prifit_bn_bwd;

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

/* Column permutation / zero padding of a weight matrix into the internal row layout of a first layer (upstream concatenates
 * [xyz | feat] or [points1 | interpolated], models/pointnet_util.py:195-197, :306; the build's rows are [feat | xyz | 0-pad] and
 * [interpolated | points1 | 0-pad], 16-byte rows): out [rows, dst_cols][r][j] = w[r][map[j]] (map[j] < 0: 0), w [rows, src_cols]. */
int prifit_pack_cols(const float *w, int rows, int src_cols, const int32_t *map, int dst_cols, float *out, void *stream);
/* Its autograd: gw [rows, src_cols][r][c] = sum of g[r][j] over the output columns j with map[j] == c, given as the CSR
 * (inv_off [src_cols + 1], inv_idx) of the map's inverse. */
int prifit_unpack_cols(const float *g, int rows, int dst_cols, const int32_t *inv_off, const int32_t *inv_idx, int src_cols,
                       float *gw, void *stream);
/* The same for up to prifit_pack_cols_max_jobs() matrices in ONE launch per direction (every first-layer weight of a network
 * at the top of its forward, their gradients at the end of its backward): HOST arrays of njobs entries each. */
int prifit_pack_cols_max_jobs(void);
int prifit_pack_cols_multi(int njobs, const float *const *w, const int32_t *rows, const int32_t *src_cols, const int32_t *const *map,
                           const int32_t *dst_cols, float *const *out, void *stream);
int prifit_unpack_cols_multi(int njobs, const float *const *g, const int32_t *rows, const int32_t *dst_cols, const int32_t *const *inv_off,
                             const int32_t *const *inv_idx, const int32_t *src_cols, float *const *gw, void *stream);

/* out[(b,n), col0 + c] = sum_j weight[b,n,j] * points2[b, idx[b,n,j], c]
 * (models/pointnet_util.py:298).  points2 [B,S,C], out rows have stride ld_out. */
int prifit_three_interpolate(const float *points2, const int32_t *idx, const float *weight, int B,
                             int N, int S, int C, int ld_out, int col0, float *out, void *stream);

/* Backward of prifit_three_interpolate w.r.t. points2 (dpoints2 [B,S,C] initialised by the caller). */
int prifit_three_interpolate_bwd(const float *gout, int ld_gout, int col0, const int32_t *idx,
                                 const float *weight, int B, int N, int S, int C, float *dpoints2,
                                 void *stream);
/* The input rows of a feature-propagation MLP in one launch (models/pointnet_util.py:287-306): out [B N, kp] =
 * [interpolated (D2) | points1 (D1) | zeros], interpolated as prifit_three_interpolate; idx == NULL: S == 1, every point takes
 * points2[b, 0] (:287-288).  points1 [B,N,D1] (NULL when D1 == 0).  Backward: prifit_three_interpolate_bwd on the first D2
 * columns (ld = kp); the points1 columns are their own gradient. */
int prifit_fp_rows(const float *points2, const int32_t *idx, const float *weight, const float *points1, int B, int N, int S, int D2,
                   int D1, int kp, float *out, void *stream);

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
#define PRIFIT_EPI_MSBWD 3    /* C = acc * aux / b^2 where aux > exp(-13), else 0: autograd of the
                                 line above (aux = the forward kernel matrix, indexed like C)     */

/* Batched GEMM with fused prologue/epilogue.  Replaces every conv1x1 of the shared per-position
 * MLPs (models/pointnet_util.py:195-199, :252-256, :310-313; models/pointnet2_part_seg_msg.py:88,
 * :109,:128) and their autograd, and the N x N x D products of src/mean_shift.py:65,73,154,168,185.
 *   lda/ldb/ldc: row strides; strideA/B/C: batch strides (elements); grid z = batch * splitk.
 *   a_scale/a_shift (both or neither): operand A is read as max(a*scale[c]+shift[c], 0) with c the
 *     index along A's contiguous dimension (k for NT/NN, m for TN) -- the train-mode BatchNorm+ReLU
 *     of the producing layer applied on load.  b_scale/b_shift: the same for B (c = k for NT, n else).
 *   bias [N] (element z*bias_batch_stride + n for batch z) or NULL.  col_stats [ceil(M/tile_m)][2][N] or NULL: per-M-tile partial column sums
 *     and sums of squares of the stored C (batch == 1, splitk == 1 only).
 *   accumulate != 0: C += result (C initialised by the caller).  splitk > 1: the K range is split
 *     over workgroups which add with float atomics (needs accumulate != 0 and PRIFIT_EPI_NONE).
 *   epi_aux / ld_aux / stride_aux: second matrix read by PRIFIT_EPI_MSBWD; epi_row_add [batch][M] (or
 *     NULL) is added to every element of its row before that transform.
 *   a_rowsum [batch][M] (or NULL; NT/NN, splitk == 1): receives sum_k A[m][k] of the (prologue-
 *     transformed) A operand -- the kernel row sums of src/mean_shift.py:70 for free.
 * 16-byte vector loads are used when pointers, strides and contiguous extents are multiples of 4
 * floats; any other shape takes a scalar-load path. */
int prifit_gemm_f32(int layout, int M, int N, int K, const float *A, long long lda, long long strideA,
                    const float *B, long long ldb, long long strideB, float *C, long long ldc,
                    long long strideC, int batch, int splitk, const float *a_scale,
                    const float *a_shift, const float *b_scale, const float *b_shift,
                    const float *bias, long long bias_batch_stride, float *col_stats, int epilogue,
                    const float *epi_batch_scalar, const float *epi_aux, long long ld_aux,
                    long long stride_aux, const float *epi_row_add, float *a_rowsum, int accumulate,
                    const prifit_bn_fwd *bn, void *stream);

/* Chord-distance matrix of a point set with itself (src/mean_shift.py:154 bandwidth statistic, :185 non-maximum
 * suppression): C[z] = 2 - 2 A[z] A[z]^T for unit rows, [n, n] per batch item; n % 128 == 0, K % 32 == 0, 16-byte rows.
 * Only the tiles on and above the diagonal are computed, the others written as their transposes: the same bits as
 * prifit_gemm_f32(NT, A, A, epilogue = chord).
 * owner_key (may be NULL): [batch, n] 64-bit keys, ALL BITS SET on entry; on return the low word of key j is
 * argmin_i C[i][j] (first minimum) and the high word the order-preserving image of that distance -- nms's owner pass
 * (src/mean_shift.py:168-170) without a second read of the matrix; hand it to prifit_nms. */
int prifit_chord_sym_f32(const float *A, long long lda, long long strideA, float *C, long long ldc, long long strideC,
                         int n, int K, int batch, unsigned long long *owner_key, void *stream);
/* The same product for nms alone (src/mean_shift.py:185-190): with the owner pass fused in (owner_key, REQUIRED by
 * prifit_nms_mask) all nms reads of the matrix is `dist[u][j] < b` -- one bit.  The matrix is not written: mask [batch, n, n / 32]
 * uint32, bit (col % 32) of word col / 32 of row `row` = (2 - 2 a_row . a_col < thr[z]) -- the comparison prifit_nms makes, on the
 * same float.  thr [batch] (the bandwidths). */
int prifit_chord_sym_mask(const float *A, long long lda, long long strideA, const float *thr, uint32_t *mask, int n, int K, int batch,
                          unsigned long long *owner_key, void *stream);
/* The plain product C[z] = A[z] A[z]^T on the same kernel (the `inner` of src/dgcnn.py:13 for a neighbour graph over features,
 * K = 64): the same bits as prifit_gemm_f32(NT, A, A), 136 of 256 tiles computed at n = 2048. */
int prifit_gram_sym_f32(const float *A, long long lda, long long strideA, float *C, long long ldc, long long strideC, int n, int K,
                        int batch, void *stream);

/* Forward of a max-pooled last layer on the tiled (persistent 128 x 128) kernel, as prifit_gemm_stream_pool_f32 does on
 * the streaming shapes (models/pointnet_util.py:252-257): Y = relu(bn(A)) W^T + bias with the column statistics AND the
 * pool candidates cand[M / 32][4][N] (max, argmax, min, argmin per 32 rows and column) for
 * prifit_pool_from_candidates.  prifit_gemm_pool_supported(M, N, K): 1 when the shape is taken (M % 32 == 0, N > 96, more
 * than 512 output tiles, 16-byte rows).  a_scale / a_shift both NULL: no prologue on A (the DGCNN global-feature layer,
 * src/dgcnn.py:194-197, whose input is already activated).  Y == NULL: the product is not stored (statistics and candidates
 * only: 201 MB less written for that layer at B = 24 x 2048). */
int prifit_gemm_pool_f32(int M, int N, int K, const float *A, long long lda, const float *W, long long ldb, float *Y,
                         long long ldc, const float *a_scale, const float *a_shift, const float *bias, float *col_stats,
                         float *cand, const prifit_bn_fwd *bn, void *stream);
int prifit_gemm_pool_supported(int M, int N, int K);

/* Weights-stationary streaming variant for the tall-and-skinny (HBM-bound) layers of the shared MLPs: layout
 * PRIFIT_GEMM_NT (C = A B^T, B [N,K]) or PRIFIT_GEMM_NN (C = A B, B [K,N]) with M >= 32768, N in {64,96,128},
 * K in {64,96,128}; A [M,K] with lda % 4 == 0.  Persistent workgroups keep B in registers and stream 64-row tiles
 * of A; same prologue (a_scale/a_shift [K]) and bias [N] semantics as prifit_gemm_f32; col_stats
 * [prifit_gemm_stream_slabs(M,K)][2][N] or NULL receives one column (sum, sum of squares) slab per workgroup. */
int prifit_gemm_stream_f32(int layout, int M, int N, int K, const float *A, long long lda, const float *B,
                           long long ldb, float *C, long long ldc, const float *a_scale, const float *a_shift,
                           const float *bias, float *col_stats, const prifit_bn_fwd *bn, void *stream);
int prifit_gemm_stream_supported(int layout, int M, int N, int K);  /* 1 / 0 */
int prifit_gemm_stream_slabs(int M, int K);                         /* workgroups = statistics slabs */

/* C (+)= A1 B1 + A2 B2 as ONE batched PRIFIT_GEMM_NN product over K1 + K2 (K1 % 32 == 0; both pairs share lda / ldb
 * and the batch strides; A1/A2 and B1/B2 equally aligned): the two dX terms of a mean-shift backward iteration,
 * dX += gS^T Z + K^T gO (autograd of src/mean_shift.py:65-73), with one epilogue instead of two. */
int prifit_gemm_dual_nn_f32(int M, int N, int K1, int K2, const float *A1, const float *A2, long long lda,
                            long long strideA, const float *B1, const float *B2, long long ldb, long long strideB,
                            float *C, long long ldc, long long strideC, int batch, int splitk, int accumulate,
                            void *stream);

/* The same fusion on the tiled kernel (any M, N, K; every dA product that is not a streaming shape): G = dY . W with
 * red_slab [ceil(M / prifit_gemm_stats_tile_m(M, N))][2][N] receiving the (m1, m2) partials described below. */
int prifit_gemm_dgrad_bnred_f32(int M, int N, int K, const float *dY, long long lda, const float *W, long long ldb,
                                float *G, long long ldc, const float *Yprev, long long ldy, const float *scale,
                                const float *shift, const float *mean, const float *invstd, float *red_slab,
                                const prifit_bn_bwd *bn, void *stream);
/* The winners' terms of a layer max-pooled over the WHOLE cloud (src/dgcnn.py:194-197: x.max(dim=-1) behind conv + GroupNorm +
 * ReLU; one pooling group per sample, K rows, Cout winners), backward in the algebraic form (csrc/pool_alg.hip):
 *   dX[b, arg[b,c], :] += T[b,c] W[c, :]   (rows; channels in ascending order)      dX [Bs K, lddx], NULL: skipped
 *   dW[c, :] += sum_b T[b,c] X[b K + arg[b,c], :]   (samples in ascending order)    dW [Cout, lddw], NULL: skipped
 * arg [Bs, Cout] the winning row inside the sample, T [Bs, Cout] the pooled gradient through the activation (0: no term).  The
 * dense terms (X_b M_b, Gram matrices) are batched prifit_gemm_f32 products on the caller's side.  Cout % 64 == 0, <= 1024; Cin in
 * {64, 128, 256}.  Deterministic. */
/* out[i] = sum over the nslab slabs of part[slab][i], i < n: per-workgroup partial results (a weight gradient's partial
 * [Cout, Cin] blocks) added in a fixed order, four interleaved chains -- the same bits from run to run. */
int prifit_slab_sum(const float *part, int nslab, long long n, float *out, void *stream);
int prifit_global_pool_winners_supported(int Cout, int Cin);
int prifit_global_pool_winners_f32(int Bs, int K, int Cout, int Cin, const int32_t *arg, const float *T, const float *W,
                                   long long ldw, const float *X, long long ldx, float *dX, long long lddx, float *dW,
                                   long long lddw, void *stream);

/* Rows that are never stored (round 4).  The first layer of a set-abstraction MLP written by linearity --
 * y1[row] = U[b, idx[row]] - Vc[b, s], U [B,N,64] per point (bias folded in), Vc [B,S,64] per centre, models/pointnet_util.py
 * :243-252 -- has ~10^6 rows of 256 bytes that its three consumers used to read back from HBM (0.6 GB written + 3 reads per
 * step for SA1); U itself is 12 MB and stays in L2 / MALL.  These entry points re-form the rows on load instead:
 * prifit_gemm_stream_gather_f32      = prifit_gemm_stream_f32 (NT, K = 64, prologue relu(a_scale y1 + a_shift)) with A = y1;
 * prifit_gemm_stream_bwd_gather_f32  = prifit_gemm_stream_bwd_f32 (Cin = 64, middle layer) with Yp = y1;
 * prifit_sa_first_layer_dw_bn_gather = prifit_sa_first_layer_dw_bn with Y1 = y1.
 * idx [M]: the index lists of prifit_sa_group_linear_fwd (which, in gather mode, takes Y[r] = NULL: lists + statistics
 * only); rows_per_centre % 64 == 0, M % (n_centres * rows_per_centre) == 0. */
int prifit_gemm_stream_gather_f32(int M, int N, const int32_t *idx, const float *U, const float *Vc, int n_points, int n_centres,
                                  int rows_per_centre, const float *B, long long ldb, float *C, long long ldc,
                                  const float *a_scale, const float *a_shift, const float *bias, float *col_stats, const prifit_bn_fwd *bn, void *stream);
int prifit_gemm_stream_bwd_gather_f32(long long P, int Cout, const float *G, const float *Y, const float *scale,
                                      const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                                      const float *W, long long ldw, const int32_t *idx, const float *U, const float *Vc,
                                      int n_points, int n_centres, int rows_per_centre, const float *p_scale,
                                      const float *p_shift, const float *p_mean, const float *p_invstd, float *Gp, long long ldgp,
                                      float *red_slab, float *dW, long long lddw, float *workspace, const prifit_bn_bwd *bn, void *stream);
/* Forward of a max-pooled last layer without re-reading it for the pool (models/pointnet_util.py:199,256):
 * prifit_gemm_stream_f32 (NT, prologue required, M % 32 == 0) that also emits, per 32-row block and column, the largest
 * and smallest stored C and the row of their first occurrence, cand [M/32][4][N]; once the BatchNorm affine (scale,
 * shift) of this layer is final, prifit_pool_from_candidates gives out [G, ldo] = max_k relu(bn(Y)) and arg [G, C] (the
 * winning sample, first maximum) for groups of K rows, K % 32 == 0 -- what prifit_pool_fwd computes from Y itself.
 * rows_per_sample > 0 (a multiple of K): scale / shift are per-sample tables [Bs][C] (GroupNorm), else one row [C].
 * ystar (may be NULL) [G, C]: the winner's value of C itself (the pre-activation under the pool) -- all a backward needs of
 * Y when the layer's gradients come from its input (csrc/pool_alg.hip), so that the product need not store Y at all. */
int prifit_gemm_stream_pool_f32(int M, int N, int K, const float *A, long long lda, const float *B, long long ldb,
                                float *C, long long ldc, const float *a_scale, const float *a_shift, const float *bias,
                                float *col_stats, float *cand, const prifit_bn_fwd *bn, void *stream);
int prifit_pool_from_candidates(const float *cand, const float *scale, const float *shift, int G, int K, int C,
                                int rows_per_sample, float slope, float *out, long long ldo, int32_t *arg, float *ystar,
                                void *stream);

/* dA of a shared-MLP layer with the BatchNorm-backward reduction of the layer below fused into the epilogue:
 * G [M,N] = dY [M,K] . W [K,N] (PRIFIT_GEMM_NN, streaming shapes), and red_slab [prifit_gemm_stream_slabs(M,K)][2][N]
 * receives per-workgroup partials of m1 = sum_rows(G * mask), m2 = sum_rows(G * mask * yhat) with
 * mask = (Yprev * scale + shift > 0), yhat = (Yprev - mean) * invstd -- the inputs of prifit_bn_bwd_finalize that
 * prifit_bn_relu_bwd_reduce would otherwise compute by re-reading G (autograd of models/pointnet_util.py:195-199). */
int prifit_gemm_stream_dgrad_f32(int M, int N, int K, const float *dY, long long lda, const float *W, long long ldb,
                                 float *G, long long ldc, const float *Yprev, long long ldy, const float *scale,
                                 const float *shift, const float *mean, const float *invstd, float *red_slab,
                                 const prifit_bn_bwd *bn, void *stream);

/* The max-pooled LAST layer of a per-group MLP (models/pointnet_util.py:199,256), autograd without materialising
 * its dY: with (a, b, d) = prifit_bn_bwd_finalize's coefficients, dY[g,k,c] = T[g,c]*[k == arg[g,c]] + b[c]*Y[g,k,c] + d[c]
 * where T[g,c] = a[c] * relu'(.) * gp[g,c] (prifit_pool_bwd_table: gp [G,C] the gradient of the pooled output, arg the
 * winners saved by prifit_pool_fwd).  The two consumers form dY while staging their operand from Y itself:
 *   prifit_gemm_stream_dgrad_pool_f32: G = dY . W for the layer below (bias_dW = d^T W [N] supplied by the caller;
 *     pool_K a multiple of 64 dividing M; red_slab / Yprev / ... as in prifit_gemm_stream_dgrad_f32, or all NULL),
 *   prifit_gemm_stream_tn_pool_f32: out += dY^T relu(bn(A)) (pool_K a multiple of 8 dividing P).
 * prifit_pool_bwd_apply (which writes dY) stays for the shapes outside the streaming kernels. */
int prifit_pool_bwd_table(const float *gp, long long ldgp, const float *Y, long long ldy, const int32_t *arg,
                          const float *scale, const float *shift, const float *coef_a, int G, int K, int C, float slope,
                          float *T, void *stream);
int prifit_gemm_stream_dgrad_pool_f32(int M, int N, int K, const float *Y, long long lda, const float *W, long long ldb,
                                      float *G, long long ldc, const float *bias_dW, const int32_t *pool_arg,
                                      const float *pool_T, const float *coef_b, int pool_K, const float *Yprev,
                                      long long ldy, const float *scale, const float *shift, const float *mean,
                                      const float *invstd, float *red_slab, const prifit_bn_bwd *bn, void *stream);
int prifit_gemm_stream_tn_pool_f32(int Mo, int No, long long P, const float *Y, long long ldy, const float *A,
                                   long long lda, float *out, long long ldo, const float *b_scale,
                                   const float *b_shift, const int32_t *pool_arg, const float *pool_T,
                                   const float *coef_b, const float *coef_d, int pool_K, float *workspace,
                                   void *stream);

/* dW = dY^T relu(bn(A)) of the same layers (PRIFIT_GEMM_TN with the reduction over P >= 32768 grouped samples,
 * P % 8 == 0, and an output of Mo x No, both multiples of 32 in 32..128): out [Mo, ldo] += sum_rows G[row,0:Mo]^T
 * A[row,0:No] with A read as max(a*b_scale[n]+b_shift[n], 0) when the prologue is given.  `out` is initialised by
 * the caller (zeros, or a gradient to accumulate into).  Every wave streams its own rows as MFMA fragments (no
 * LDS staging); per-workgroup partial slabs go to `workspace` (prifit_gemm_stream_tn_workspace(Mo,No,P) floats,
 * caller-owned scratch) and a second small launch adds them to `out`. */
/* The same two streaming products for a MIDDLE layer of a shared MLP (autograd of models/pointnet_util.py:195-199 /
 * :252-256): the layer's dY = a*(Y*scale+shift > 0 ? G : 0) + b*Y + d -- prifit_bn_relu_bwd_apply's expression with
 * the coefficients of prifit_bn_bwd_finalize -- is formed from G (gradient w.r.t. relu(bn(Y)), leading dimension ldy)
 * and the pre-activation Y inside the kernels: no pass writing dY.
 *   prifit_gemm_stream_tn_bn_f32:    out += dY^T relu(bn(A))
 *   prifit_gemm_stream_dgrad_bn_f32: Gprev = dY W, with the BatchNorm-backward partials of the previous layer in
 *                                    red_slab (as prifit_gemm_stream_dgrad_f32) */
int prifit_gemm_stream_tn_bn_f32(int Mo, int No, long long P, const float *G, const float *Y, long long ldy, const float *A,
                                 long long lda, float *out, long long ldo, const float *b_scale, const float *b_shift,
                                 const float *scale, const float *shift, const float *coef_a, const float *coef_b,
                                 const float *coef_d, float *workspace, void *stream);
int prifit_gemm_stream_dgrad_bn_f32(int M, int N, int K, const float *Gin, const float *Y, long long lda, const float *W,
                                    long long ldb, float *G, long long ldc, const float *scale_l, const float *shift_l,
                                    const float *coef_a, const float *coef_b, const float *coef_d, const float *Yprev,
                                    long long ldy, const float *scale, const float *shift, const float *mean,
                                    const float *invstd, float *red_slab, const prifit_bn_bwd *bn, void *stream);
int prifit_gemm_stream_tn_f32(int Mo, int No, long long P, const float *G, long long ldg, const float *A,
                              long long lda, float *out, long long ldo, const float *b_scale,
                              const float *b_shift, float *workspace, void *stream);
long long prifit_gemm_stream_tn_workspace(int Mo, int No, long long P);  /* floats */
int prifit_gemm_stream_tn_supported(int Mo, int No, long long P);  /* 1 / 0 */

/* Rows of C covered by one col_stats slab of prifit_gemm_f32 (its M tile). */
int prifit_gemm_tile_m(int N);
/* Rows per col_stats slab for a batch == 1, splitk == 1 product of M x N: 128, or 64 when the product has so few
 * output tiles that prifit_gemm_f32 switches to 64 x 64 tiles. */
int prifit_gemm_stats_tile_m(int M, int N);

/* ------------------------------------------------------------------------------------------ */
/* train-mode BatchNorm + ReLU + group max-pool around the GEMMs                                */
/* (matrices are [P, ld] channels-last, C % 4 == 0, rows 16-byte aligned)                       */
/* ------------------------------------------------------------------------------------------ */

/* Rows reduced into one partial slab by the *_reduce / col_stats kernels below. */
int prifit_reduce_rows_per_slab(void);
/* groups (rows of the pooled gradient) per slab of prifit_pool_bwd_reduce: its slab is [ceil(G/this)][2][C] */
int prifit_pool_reduce_groups_per_slab(void);

/* slab [ceil(P/rows_per_slab)][2][C] = per-block column (sum, sum of squares) of Y. */
int prifit_col_stats(const float *Y, long long ld, int P, int C, float *slab, void *stream);
/* out [C] = column sums of Y [P, ld] (C % 4 == 0, 16-byte rows): the bias gradient of a 1x1 convolution that has no BatchNorm
 * behind it (autograd of conv2 / extra_conv_emb, models/pointnet2_part_seg_msg.py:109,128, and of the DGCNN decoder's biased
 * convolutions, src/dgcnn.py:236-259).  Per-workgroup partial rows in `workspace` (prifit_col_sum_workspace floats, 16-byte
 * aligned), added in a fixed order by a second small launch: no atomics, the same bits from run to run. */
long long prifit_col_sum_workspace(int P, int C);
int prifit_col_sum(const float *Y, long long ld, int P, int C, float *out, float *workspace, void *stream);
/* The same per sample: out [P / rows_per_sample, C] = the column sums of each sample's rows (rows_per_sample % 512 == 0; the
 * 1^T x_b of a per-sample rank-one term, csrc/pool_alg.hip's global-pool backward). */
int prifit_col_sum_samples(const float *Y, long long ld, int P, int C, int rows_per_sample, float *out, float *workspace,
                           void *stream);

/* GroupNorm statistics (nn.GroupNorm of src/dgcnn.py:150-171,203-213: per sample and channel group) -> the same affine
 * form, per-sample tables [Bs][C].  slab [Bs * slabs_per_sample][2][C]: column (sum, sum of squares) partials, consecutive
 * slabs of one sample; count = rows per sample x channels per group; fp64 accumulation.  C / groups divides 256
 * (prifit_gn_finalize_supported).  _bwd_: from the (sum Gm, sum Gm yhat) partials of the backward reduce kernels the apply
 * pass's coefficients coef_b, coef_d [Bs][C] (coef_a = scale) and the per-sample channel totals S [Bs][2][C] in fp64
 * (dgamma = sum_b S[b][1], dbeta = sum_b S[b][0]).
 * chsum (forward: optional output, backward: optional input) [Bs][C] fp64 = the column sums of Y per sample.  With it the
 * backward also writes dsum [Bs][C] = the column sums of dY per sample -- from dY = coef_a Gm + coef_b Y + coef_d row by row,
 * sum_rows dY = coef_a sum Gm + coef_b sum Y + coef_d rows_per_sample -- i.e. the gradient of a per-sample offset
 * (prifit_gn_finalize_offset) and, summed over the samples (prifit_gn_param_grads), of the bias of the convolution in front
 * (nn.Conv1d(.., bias=True), src/dgcnn.py:188,236-240), without a pass over the [B N, C] tensor dY. */
int prifit_gn_finalize_supported(int C, int groups);
int prifit_gn_finalize(const float *slab, int Bs, int slabs_per_sample, int C, int groups, double count, const float *gamma,
                       const float *beta, double eps, float *scale, float *shift, float *mean, float *invstd, double *chsum,
                       void *stream);
int prifit_gn_bwd_finalize(const float *slab, int Bs, int slabs_per_sample, int C, int groups, double count,
                           const float *gamma, const float *mean, const float *invstd, float *coef_b, float *coef_d,
                           double *S, const double *chsum, double rows_per_sample, float *dsum, void *stream);
/* The same for a tensor Y + offset[b][c] whose per-sample, per-channel constant `offset` [Bs][C] the producer left out of
 * Y (the decoder's first convolution, src/dgcnn.py:253-257: the 1024 global-feature channels of its input are the same for
 * every point of a sample, so their product with the weight is ONE row per sample instead of N): slab = statistics of Y
 * alone, rows_per_sample = N.  The tables are relative to Y (mean' = mean - offset, shift' = beta - mean' scale): the
 * affine / pool / backward kernels then run on Y unchanged and the offset never has to be added to N rows. */
int prifit_gn_finalize_offset(const float *slab, int Bs, int slabs_per_sample, int C, int groups, double count,
                              const float *gamma, const float *beta, double eps, const float *offset, double rows_per_sample,
                              float *scale, float *shift, float *mean, float *invstd, double *chsum, void *stream);
/* dgamma [C], dbeta [C] = the sums over the samples of S[b][1][:], S[b][0][:] (src/dgcnn.py:150-171: the affine parameters of
 * nn.GroupNorm are shared by all samples); dsum (may be NULL) [Bs][C] from prifit_gn_bwd_finalize: db [C] = its sum over the
 * samples. */
int prifit_gn_param_grads(const double *S, int Bs, int C, float *dgamma, float *dbeta, const float *dsum, float *db,
                          void *stream);

/* Batch statistics -> affine form of BatchNorm (torch.nn.BatchNorm{1,2}d in train mode, as used at
 * models/pointnet_util.py:198,254,312): mean/var over `count` positions from the partial slabs,
 * scale = gamma*invstd, shift = beta - mean*scale; running stats (may be NULL) updated with
 * `momentum` and the unbiased variance. */
int prifit_bn_finalize(const float *slab, int nslab, int C, double count, const float *gamma,
                       const float *beta, float eps, float momentum, float *running_mean,
                       float *running_var, float *scale, float *shift, float *mean, float *invstd,
                       void *stream);

/* The kernels below take the normalisation as per-channel tables.  rows_per_sample == 0: ONE table row
 * [C] for all rows (BatchNorm).  rows_per_sample > 0: tables are [P/rows_per_sample][C] and row r uses
 * table row r / rows_per_sample (GroupNorm of src/dgcnn.py:157-159: statistics per sample).
 * slope: negative slope of the activation (0 = ReLU, 0.2 = the LeakyReLU of src/dgcnn.py:162). */

/* out = act(Y*scale + shift): F.relu(bn(.)) materialised (module outputs). */
int prifit_affine_relu(const float *Y, long long ldy, const float *scale, const float *shift, int P,
                       int C, int rows_per_sample, float slope, float *out, long long ldo, void *stream);

/* torch.max(relu(bn(Y)), dim=K)[0] (models/pointnet_util.py:199,256): Y [G*K, ldy] -> out [G, ldo],
 * arg [G, C] = index k of the first maximum. */
int prifit_pool_fwd(const float *Y, long long ldy, const float *scale, const float *shift, int G, int K,
                    int C, int rows_per_sample, float slope, float *out, long long ldo, int32_t *arg,
                    void *stream);

/* First layer of a set-abstraction MLP by linearity (models/pointnet_util.py:243-252 / :127-133,195-197):
 * conv1([feat_j | xyz_j - c_g]) = U_j - Vc_g + bias with U [B,N,C] = [feat | xyz] W1^T per point and
 * Vc [B,S,C] = c W1x^T per centre: Y [(b,s,k), C] = U[b, idx[b,s,k]] - Vc[b,s] + bias (bias may be NULL), plus
 * per-slab column (sum, sum of squares) [ceil(P/rows)][2][C], rows = prifit_reduce_rows_per_slab(), for the
 * following BatchNorm. */
int prifit_gather_linear_fwd(const float *U, const float *Vc, const float *bias, const int32_t *idx, int B,
                             int N, int S, int K, int C, float *Y, float *slab, void *stream);
/* autograd: dU [B,N,C] (initialised by the caller) += scatter of dY; dVc [B,S,C] = -sum_k dY. */
int prifit_gather_linear_bwd(const float *dY, const int32_t *idx, int B, int N, int S, int K, int C,
                             float *dU, float *dVc, void *stream);
/* Backward of a max-pooled, normalised by-linearity layer in one pass (the DGCNN edge convolution of src/dgcnn.py:98-107,
 * :157-171 written as y[(g,k)] = U[idx[g,k]] - Vc[g] -> GroupNorm -> LeakyReLU -> max over the K neighbours): with gp [G, C]
 * the gradient of the pooled output (G = B S groups), Y [G K, C] the pre-activations, arg [G, C] the winners, and the
 * coefficient tables of prifit_pool_bwd_apply (rows_per_sample > 0: per-sample rows, GroupNorm), forms
 * dy = a (k == arg ? act'(y) gp : 0) + b y + d on the fly and accumulates dU [B,N,C] += dy at idx (float atomics; zero it
 * first) and dVc [B,S,C] = -sum_k dy.  dY is never written. */
int prifit_gather_linear_bwd_pool(const float *gp, long long ldgp, const float *Y, const int32_t *arg, const float *scale,
                                  const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                                  const int32_t *idx, int B, int N, int S, int K, int C, int rows_per_sample, float slope,
                                  float *dU, float *dVc, void *stream);

/* ---- The DGCNN edge convolution block without any per-edge tensor (csrc/edge_conv.hip; replaces, for src/dgcnn.py:98-107
 * get_graph_feature + :157-171 conv -> GroupNorm -> LeakyReLU -> max over the k neighbours, the pair prifit_gather_linear_fwd /
 * prifit_gather_linear_bwd_pool and their [B N k, C] pre-activations).  y[(i,j)] = U[b, idx[b,i,j]] - Vc[b,i]; U, Vc [B,N,C],
 * idx [B,N,k] int32 (an entry outside [0,N) is a zero row without gradient), C in {64, 128, 256}, N % 16 == 0, N <= 8192. */
int prifit_edge_tables_supported(int N, int k, int C);
/* points per GroupNorm statistics slab of prifit_edge_stats (slabs per sample = N / this) */
int prifit_edge_points_per_slab(void);
/* offs [B][N+1], lst [B][N k]: for every point n of shape b the edges i k + j with idx[b,i,j] == n are lst[b][offs[b][n] ..
 * offs[b][n+1]) (the order inside a list is not defined); pos [B][N k]: the position of edge i k + j in lst[b] (-1 for an
 * entry outside [0,N)).  One graph serves every layer that uses it. */
int prifit_edge_csr(const int32_t *idx, int B, int N, int k, int32_t *offs, int32_t *lst, int32_t *pos, void *stream);
/* One pass over the neighbour lists.  U [B N, ldu] and Vc [B N, ldv] rows; selfterm = 0: y[(i,j)] = U[idx[i,j]] - Vc[i];
 * selfterm = 1: Vc holds Vb = X Wb^T and the centre term is U[i] - Vb[i] (W [x_j - x_i | x_i] = U_j - U_i + Vb_i: U and Vb are
 * then the two halves of ONE product X [Wa; Wb]^T, ldu = ldv = 2 C).  Per (point, channel): ymax / ymin = max / min over j of
 * y, karg = first position of the maximum | first position of the minimum << 10 | number of valid neighbours << 20 (k <
 * 1024), ysum = sum over j of y, vct = the centre term [B N, C] (kept for the backward); slab [B N/P][2][C], P =
 * prifit_edge_points_per_slab() = column sum / sum of squares of y over each P points x k rows (prifit_gn_finalize's layout). */
int prifit_edge_stats(const float *U, long long ldu, const float *Vc, long long ldv, int selfterm, const int32_t *idx, int B, int N,
                      int k, int C, float *ymax, float *ymin, int32_t *karg, float *ysum, float *vct, float *slab, void *stream);
/* With the per-sample tables scale / shift [B,C] of the finalized statistics: out [B N, ldo] = act(scale y* + shift), the
 * maximum over the k neighbours of the activation (y* = ymax where scale >= 0, ymin where scale < 0: the activation is
 * monotone in y), and ystar [B N, C] = y* for the backward. */
int prifit_edge_pool(const float *ymax, const float *ymin, const float *scale, const float *shift, int B, int N, int C,
                     float slope, float *out, long long ldo, float *ystar, void *stream);
/* Backward with the coefficient tables [B,C] of the pooled GroupNorm backward (dy = a T [j == winner] + b y + d, T =
 * act'(scale y* + shift) gp), vct from prifit_edge_stats: the gradient of the centre term dVc [B N, ldd] = -(a T + b ysum +
 * (valid neighbours) d); dU [B N, ldd] (written, not accumulated) = b (deg U - sum of vct over the in-edges) + d deg + sum of
 * a T over the in-edges that won, all from the CSR with fp64 sums: no atomics.  selfterm = 1: the second output is the
 * gradient of Vb (= -dVc) and dU also receives the point's own dVc (dU and dVb are then the halves of one [B N, 2 C] gradient,
 * ldd = 2 C).  workspace: prifit_edge_bwd_workspace bytes, 8-byte aligned. */
long long prifit_edge_bwd_workspace(int B, int N, int k, int C);
int prifit_edge_bwd(const float *gp, long long ldgp, const float *ystar, const float *ysum, const int32_t *karg,
                    const float *scale, const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                    const float *U, long long ldu, const float *vct, int selfterm, const int32_t *idx, const int32_t *offs,
                    const int32_t *lst, const int32_t *pos, int B, int N, int k, int C, float slope, float *dU, float *dVc,
                    long long ldd, void *workspace, void *stream);

/* The same autograd with the train-mode BatchNorm + ReLU backward of the gathered layer folded in (what
 * prifit_bn_relu_bwd_apply would have written first): dY = a (Y s + t > 0 ? G : 0) + (b Y + d) is formed on load from G
 * (gradient w.r.t. the layer's ReLU output) and Y (its pre-activation), [B*S*K, C] each; per-channel scale / shift and
 * the coefficients a, b, d of prifit_bn_bwd_finalize.  dU [B,N,C] AND dVc [B,S,C] arrive ZERO-INITIALISED and are
 * accumulated into (scatter staged in LDS per shape and range of 128 points; C <= 128, even --
 * prifit_gather_linear_bwd_bn_supported).  Replaces autograd of models/pointnet_util.py:243-252 through the first conv + BatchNorm. */
int prifit_gather_linear_bwd_bn_supported(int N, int C);
int prifit_gather_linear_bwd_bn(const float *G, const float *Y, const float *scale, const float *shift, const float *coef_a,
                                const float *coef_b, const float *coef_d, const int32_t *idx, int B, int N, int S, int K,
                                int C, float *dU, float *dVc, void *stream);
/* The same gradients as a GATHER (round 5): no atomics, no LDS staging, and the layer's rows y1 are not read -- all in-edges of
 * point n carry the same row U[n], so y1[(g, k)] = (U[b, n] - Vc[b, g]) + bias is re-formed from the [S, C] table of the centres
 * (the subtraction the forward did: the same bits) and only G is read (once per pass).  offs [B, N + 1], lst [B, S K]: the CSR of
 * the index lists idx [B, S, K] over the N points with the owner of every list position (prifit_list_csr).  The lists are cut
 * into chunks of 64 entries, one wave each, whatever the in-degrees (ball queries pad with their first index: a few points own
 * thousands of entries); a point whose list spans chunks is summed from per-chunk partials in chunk order by a second small
 * pass.  dU [B,N,C] and dVc [B,S,C] are WRITTEN (no zero-init).  workspace: prifit_gather_linear_bwd_csr_workspace doubles.
 * bias may be NULL.  C <= 128, even. */
int prifit_gather_linear_bwd_csr_supported(int N, int C);
long long prifit_gather_linear_bwd_csr_workspace(int B, int S, int K, int C);
int prifit_gather_linear_bwd_csr(const float *G, const float *U, const float *Vc, const float *bias, const float *scale,
                                 const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                                 const int32_t *idx, const int32_t *offs, const int32_t *lst, const int32_t *owner, int B, int N,
                                 int S, int K, int C, float *dU, float *dVc, double *workspace, void *stream);
/* CSR of B sets of index lists: idx [B, E] with values in [0, nbins) (others: in no list) -> offs [B, nbins + 1], lst [B, E]
 * (the positions e of the entries with idx[e] = n at lst[offs[n] .. offs[n + 1])), pos [B, E] (where entry e sits, -1: nowhere),
 * owner (may be NULL) [B, E]: the bin of every list position.  prifit_edge_csr is this with E = N k.  nbins <= 8192. */
int prifit_list_csr(const int32_t *idx, int B, int nbins, int E, int32_t *offs, int32_t *lst, int32_t *pos, int32_t *owner,
                    void *stream);


/* Set-abstraction front end in ONE launch per layer: multi-radius ball query (models/pointnet_util.py:87-107 with
 * :19-40 fused, bit-exact like prifit_ball_query) + grouping (:43-60, :127-133 / :243-249) + the first 1x1 conv of
 * every per-radius MLP (:195-197 / :250-252), one wave per query centre, the shape's cloud in LDS.  N <= 2048,
 * R <= 4, sum(nsample) <= 320, width[r] a power of two in 16..128.  Host arrays of length R: radius2 (fp32 of
 * radius^2 like prifit_ball_query), nsample, width (C1 of the first layer), bias (device [C1] or NULL entries),
 * Y (device [B*S*K_r, C1_r] pre-activations), slab (device [B*ceil(S/q)][2][C1_r] column sum / sum of squares
 * for the BatchNorm that follows, q = prifit_sa_group_queries_per_slab(B,S); NULL entries skip them),
 * idx (device int32 [B,S,K_r], written).
 *   mode 0 "direct": y = W_r [feat_j | xyz_j - c] + bias from the upstream weight W_r [C1, D+3] (device pointers
 *     in the host array W), column order [feat, rel] when feat_first else [rel, feat]; feat [B,N,D], D in {0,3,6};
 *     feat_xyz != 0 promises that feature channels 0..2 are the coordinates (models/pointnet2_part_seg_msg.py:69-75:
 *     l0_points = xyz), which are then taken from the LDS copy of the cloud instead of being gathered;
 *   mode 1 "gather": y = U_r[b, j] - Vc_r[b, s] + bias (the layer by linearity, see prifit_gather_linear_fwd);
 *     U / Vc host arrays of device pointers [B,N,C1_r] / [B,S,C1_r]. */
int prifit_sa_group_linear_fwd(const float *xyz, const float *new_xyz, int B, int N, int S, int R,
                               const float *radius2, const int *nsample, const int *width, int mode,
                               const float *feat, int D, int feat_first, int feat_xyz, const float *const *W,
                               const float *const *U, const float *const *Vc, const float *const *bias,
                               float *const *Y, float *const *slab, int32_t *const *idx, const prifit_bn_fwd *const *bn, void *stream);
/* Queries per statistics slab of the call above (the workgroup size it will pick for B shapes x S centres). */
int prifit_sa_group_queries_per_slab(int B, int S);
/* autograd of mode 0 w.r.t. the weight: partial [nblocks][C][D+3] (upstream column order) with
 * sum_blocks partial = dY^T [feat_j | xyz_j - c] over all grouped samples; dY [B*S*K, C]. */
int prifit_sa_first_layer_dw(const float *dY, const int32_t *idx, const float *xyz, const float *new_xyz,
                             const float *feat, int B, int N, int S, int K, int C, int D, int feat_first,
                             int nblocks, float *partial, void *stream);

/* The same weight gradient with the BatchNorm + ReLU backward of that first layer fused in: G [B*S*K, C] is the
 * gradient w.r.t. relu(bn(Y1)), Y1 the pre-activations prifit_sa_group_linear_fwd wrote, (scale, shift) the forward
 * affine and (a, b, d) prifit_bn_bwd_finalize's coefficients: dy = a*(Y1*scale+shift > 0 ? g : 0) + b*Y1 + d is formed
 * on load, so no prifit_bn_relu_bwd_apply pass writes dY for this layer. */
int prifit_sa_first_layer_dw_bn(const float *G, const float *Y1, const float *scale, const float *shift,
                                const float *coef_a, const float *coef_b, const float *coef_d, const int32_t *idx,
                                const float *xyz, const float *new_xyz, const float *feat, int B, int N, int S, int K,
                                int C, int D, int feat_first, int nblocks, float *partial, void *stream);
/* The first layer by linearity for NARROW inputs (models/pointnet_util.py:243-252 with 3..9 data channels): the per-point
 * table U_r [B,N,C_r] = W_r [feat_n | xyz_n] + b_r and the per-centre table Vc_r [B,S,C_r] = W_r,x c_s of every radius of a
 * level in one launch (W_r [C_r][D+3] in upstream column order, MSG [feat, rel] or SSG [rel, feat]), so that the layer's
 * pre-activation of grouped sample (s, j) is U_j - Vc_s.  feat [B,N,D] or NULL (D = 0). */
int prifit_sa_point_tables(const float *xyz, const float *new_xyz, const float *feat, int B, int N, int S, int D, int feat_first,
                           int R, const int *width, const float *const *W, const float *const *bias, float *const *U,
                           float *const *Vc, void *stream);

/* The same when the first-layer rows Y1 are NOT stored (the layer written by linearity, y1[row] = U[b, idx[row]] - Vc[b, s]
 * with U [B,N,C] per point and Vc [B,S,C] per centre, bias folded into U): Y1 is re-formed on load from (idx, U, Vc). */
int prifit_sa_first_layer_dw_bn_gather(const float *G, const float *U, const float *Vc, const float *scale, const float *shift,
                                       const float *coef_a, const float *coef_b, const float *coef_d, const int32_t *idx,
                                       const float *xyz, const float *new_xyz, const float *feat, int B, int N, int S, int K,
                                       int C, int D, int feat_first, int nblocks, float *partial, void *stream);


/* Backward of relu(bn(Y)) given G = dL/d(relu output): partial slabs of m1 = sum(G*mask) and
 * m2 = sum(G*mask*yhat). */
int prifit_bn_relu_bwd_reduce(const float *G, long long ldg, const float *Y, long long ldy,
                              const float *scale, const float *shift, const float *mean,
                              const float *invstd, int P, int C, int rows_per_sample, float slope,
                              float *slab, const prifit_bn_bwd *bn, void *stream);

/* The same partials when the gradient gp [G, ldgp] arrives through the group max-pool. */
int prifit_pool_bwd_reduce(const float *gp, long long ldgp, const float *Y, long long ldy,
                           const int32_t *arg, const float *scale, const float *shift,
                           const float *mean, const float *invstd, int G, int K, int C,
                           int rows_per_sample, float slope, float *slab, const prifit_bn_bwd *bn, void *stream);

/* m1, m2 -> dgamma, dbeta and the per-channel coefficients of dY = a*(G*mask) + b*Y + d
 * (training != 0: batch-stat BatchNorm backward; training == 0: running-stat affine). */
int prifit_bn_bwd_finalize(const float *slab, int nslab, int C, double count, int training,
                           const float *scale, const float *mean, const float *invstd, float *dgamma,
                           float *dbeta, float *coef_a, float *coef_b, float *coef_d, void *stream);

int prifit_bn_relu_bwd_apply(const float *G, long long ldg, const float *Y, long long ldy,
                             const float *scale, const float *shift, const float *coef_a,
                             const float *coef_b, const float *coef_d, int P, int C,
                             int rows_per_sample, float slope, float *dY, long long ldd, void *stream);

int prifit_pool_bwd_apply(const float *gp, long long ldgp, const float *Y, long long ldy,
                          const int32_t *arg, const float *scale, const float *shift,
                          const float *coef_a, const float *coef_b, const float *coef_d, int G, int K,
                          int C, int rows_per_sample, float slope, float *dY, long long ldd,
                          void *stream);

/* ------------------------------------------------------------------------------------------ */
/* mean-shift clustering on the unit hypersphere (src/mean_shift.py)                            */
/* ------------------------------------------------------------------------------------------ */

/* Y = normalize(normalize(X)) over rows of D <= 256 floats, F.normalize semantics (x / max(|x|, eps)) applied twice
 * as convex_loss.py:41,57 does with the per-point embedding; _bwd: its autograd (GX from X and the gradient G of Y). */
int prifit_row_normalize2_fwd(const float *X, int D, long long rows, float eps, float *Y, void *stream);
int prifit_row_normalize2_bwd(const float *X, const float *G, int D, long long rows, float eps, float *GX, void *stream);

/* out[row] = k-th smallest entry (1-based) of row `row` of M [rows, C], C <= 4096: the
 * torch.topk(dist, k, largest=False)[0][:, -1] of compute_bandwidth, src/mean_shift.py:156-158. */
int prifit_kth_smallest_rows(const float *M, long long rows, int C, int k, float *out, void *stream);

/* One mean-shift update, src/mean_shift.py:70-82.  O [rows, D] = K . X, rowsum [rows] = sum_j K[i][j]
 * (prifit_gemm_f32's a_rowsum), Z [rows, D] the current points:
 * out = normalize(Z + (O / rowsum - Z)), nrm [rows] = the norm before normalisation. */
int prifit_meanshift_update_fwd(const float *O, const float *rowsum, const float *Z, int D,
                                long long rows, float *out, float *nrm, void *stream);

/* Autograd of the update: g = dL/d(out) -> gO = dL/dO (shape b at gO + b*gO_batch_stride, [N, D] rows)
 * and g_rowsum [B*N] = dL/d(rowsum). */
int prifit_meanshift_update_bwd(const float *g, const float *out, const float *nrm, const float *O,
                                const float *rowsum, int D, int B, int N, float *gO,
                                long long gO_batch_stride, float *g_rowsum, void *stream);

/* Flash-style fused mean-shift iteration for D == 128 (src/mean_shift.py:61-82): one workgroup per 64
 * points streams the dictionary X through LDS; S = Z X^T, K = exp(clamp((S-1)/b^2)) and O = K X are
 * chained on the matrix cores without leaving registers, followed by the normalisation epilogue.
 * Z, X [B,N,128]; bw [B].  Outputs: Znext [B,N,128]; saved for autograd: KT = K^T, element (b, key, query)
 * at KT[b*stride_kt + key*ld_kt + query] (may be NULL: the row-sparse backward does not need it), O [B,N,128] (may be
 * NULL), rowsum [B,N], nrm [B,N].  (A stream-K schedule of this kernel measured slower than the plain grid -- its time is
 * linear in the number of query blocks -- and is not kept; the dZ mode below keeps its own.) */
int prifit_meanshift_fused_fwd(const float *Z, const float *X, const float *bw, int B, int N, int D,
                               float *KT, long long ld_kt, long long stride_kt, float *Znext, float *O,
                               float *rowsum, float *nrm, void *stream);
/* The FIRST update of a trajectory (Z_0 = X, src/mean_shift.py:60): its score matrix X X^T is symmetric and already in HBM as the
 * chord matrix 2 - 2 X X^T the bandwidth step wrote (src/mean_shift.py:154-158; prifit_chord_sym_f32) -- the S product, half
 * of the update's matrix work, becomes a read of chord [B][N][ld_c] (s = 1 - chord / 2).  Outputs as prifit_meanshift_fused_fwd
 * without K^T.  N % 64 == 0, D == 128 (prifit_meanshift_fused_first_supported). */
int prifit_meanshift_fused_first_supported(int N, int D);
int prifit_meanshift_fused_first_fwd(const float *X, const float *chord, long long ld_c, long long stride_c, const float *bw, int B,
                                     int N, int D, float *Znext, float *O, float *rowsum, float *nrm, void *stream);
/* dZ = gS X with gS = (gO X^T + g_rowsum 1^T) * K / b^2 (clamp-masked), same fused data flow;
 * gST (may be NULL) receives gS^T in the layout of KT (for the dX GEMM).
 * balanced != 0: the caller hands in a ZERO-INITIALISED dZ and allows the stream-K schedule (a grid of exactly the
 * resident workgroup slots; query blocks that are split over two workgroups add their halves with float atomics);
 * used when the plain grid would leave a partial last round (N % 64 == 0 only). */
int prifit_meanshift_fused_bwd_dz(const float *gO, long long gO_batch_stride, const float *X,
                                  const float *bw, const float *g_rowsum, const float *KT, long long ld_kt,
                                  long long stride_kt, float *gST, int B, int N, int D, float *dZ,
                                  int balanced, void *stream);
/* dX += gS^T Z + K^T gO from the two saved streams (gS^T written by prifit_meanshift_fused_bwd_dz, K^T by
 * prifit_meanshift_fused_fwd; row-major [key][query], leading dimension ld_kt, 16-byte aligned), key-major: every lane
 * reads its own key row of both streams straight into MFMA A operands, only Z and gO tiles are staged in LDS.
 * N % 64 == 0, D == 128.  Replaces the two dX products of a mean-shift backward iteration (src/mean_shift.py:65,73
 * through autograd) -- the default; prifit_gemm_dual_nn_f32 is the general-shape fallback. */
int prifit_meanshift_dx_streams(const float *gO, const float *Z, const float *gST, const float *KT, long long ld_kt,
                                long long stride_kt, int B, int N, int D, float *dX, void *stream);
/* dX += gS^T Z + K^T gO (both uses of the dictionary in one iteration) with gS re-formed in registers,
 * key-major, N % 4 == 0.  (Alternative to one long-K GEMM on [gS^T | K^T]; kept for N where that buffer
 * would not fit.) */
int prifit_meanshift_fused_bwd_dx(const float *gO, const float *Z, const float *X, const float *bw,
                                  const float *g_rowsum, const float *KT, long long ld_kt,
                                  long long stride_kt, int B, int N, int D, float *dX, void *stream);

/* Backward of one (conv1x1 + train-mode BatchNorm + ReLU) layer of a shared MLP on the tall-and-skinny shapes, BOTH
 * products in one pass over the rows (autograd of models/pointnet_util.py:195-199, :252-256; replaces a
 * prifit_gemm_stream_dgrad_{bn,pool}_f32 + prifit_gemm_stream_tn_{bn,pool}_f32 pair, which each stream the same tensors):
 *     dY = a (Y s + t > 0 ? G : 0) + (b Y + d)       middle layer (pool_arg == NULL)
 *        = b Y + d + (k == arg ? T : 0)              max-pooled last layer (pool_arg / pool_T [P / pool_K, Cout], pool_K % 64 == 0)
 *     Gp [P, Cin] = dY W;  red_slab [slabs][2][Cin] = the (m1, m2) column sums of the BatchNorm backward of the layer below
 *     (pre-activation Yp [P, Cin], its scale / shift / mean / invstd);  dW [Cout, Cin] = dY^T relu(p_scale Yp + p_shift).
 * G, Y [P, Cout] contiguous; W [Cout, Cin] (ldw); (Cout, Cin) in {(128,128), (128,96), (128,64), (96,64), (64,64)}
 * (prifit_gemm_stream_bwd_supported); slabs = prifit_gemm_stream_bwd_slabs; workspace: prifit_gemm_stream_bwd_workspace
 * floats.  Deterministic (per-workgroup slabs, fixed summation order). */
int prifit_gemm_stream_bwd_supported(long long P, int Cout, int Cin, int pool_K);
int prifit_gemm_stream_bwd_slabs(long long P, int Cout, int Cin);
long long prifit_gemm_stream_bwd_workspace(long long P, int Cout, int Cin);
int prifit_gemm_stream_bwd_f32(long long P, int Cout, int Cin, const float *G, const float *Y, const float *scale,
                               const float *shift, const float *coef_a, const float *coef_b, const float *coef_d,
                               const int32_t *pool_arg, const float *pool_T, int pool_K, const float *W, long long ldw,
                               const float *Yp, long long ldyp, const float *p_scale, const float *p_shift,
                               const float *p_mean, const float *p_invstd, float *Gp, long long ldgp, float *red_slab,
                               float *dW, long long lddw, float *workspace, const prifit_bn_bwd *bn, void *stream);

/* Row-sparse backward of `iterations` mean-shift updates, for a loss that reads the shifted points through
 * `center = new_X[indices]` only (src/mean_shift.py:44-46; autograd of :61-82).  Row i of an iterate depends on row i of
 * the previous iterate alone (the dictionary is the fixed input X, :65), so d loss / d new_X is non-zero on the R kept
 * rows in EVERY iteration; the dense backward's other N - R query rows multiply exact zeros.  This entry point does the
 * R x N x D products only and re-forms the K values under the kept rows from the saved iterate rows -- the forward need
 * not keep K^T (call prifit_meanshift_fused_fwd with KT = NULL).
 * X [B,N,D], bw [B]; Zin / Zout / O / rowsum / nrm: HOST arrays of T device pointers, the tensors iteration t consumed
 * (Zin[t] [B,N,D]) and produced (Zout[t] [B,N,D], O[t] [B,N,D], rowsum[t] [B,N], nrm[t] [B,N]; Zin[t+1] == Zout[t]);
 * ids [B,R] int64 row indices (clamped to [0,N)), nrows [B] live slots per shape (NULL: all R), R <= 64;
 * g_rows [B,R,D] = dL/d(Zout[T-1][b, ids[b,r]]); dX [B,N,D] is ACCUMULATED into (both uses of the dictionary and the
 * Z_0 = X.clone() of :60) -- read-modified-written ONCE, by the last launch: the iterations leave their K and gS values in
 * tables [T][B][R][N] inside the workspace.  mode 2: TWO launches -- one workgroup per (shape, live row) runs all T iterations
 * of its row (the rows are independent of each other in the backward as in the forward; nothing crosses workgroups), then the
 * dX pass; the choice for few live rows per shape (<= ~16 at B = 24).  mode 1: THREE launches -- the T key-tiled iterations are
 * one launch that works through a queue of (iteration, shape, key tile) items, an item waiting only for items claimed before
 * it (no co-residency assumption).  mode 0: one key-tiled launch per iteration (T + 2 launches).  Modes 0 and 1 give the same
 * bits, mode 2 the same values to fp32 rounding (another summation order over the keys); T <= 16 for modes 1 and 2 (above:
 * as mode 0).  workspace: prifit_meanshift_rows_bwd_workspace(B, N, D, R, T)
 * floats, 16-byte aligned.  D in {32, 64, 128} (prifit_meanshift_rows_supported).  Deterministic (fixed summation orders;
 * two live slots that name one point are added in slot order). */
int prifit_meanshift_rows_supported(int N, int D, int R);
long long prifit_meanshift_rows_bwd_workspace(int B, int N, int D, int R, int T);
int prifit_meanshift_rows_bwd(const float *X, const float *bw, int B, int N, int D, int T, const float *const *Zin,
                              const float *const *Zout, const float *const *O, const float *const *rowsum,
                              const float *const *nrm, const long long *ids, const int *nrows, int R,
                              const float *g_rows, float *workspace, float *dX, int mode, void *stream);

/* LABELLED EXPERIMENT (default off; never the reported precision): the two matrix products of a mean-shift update,
 * S = Z X^T (src/mean_shift.py:65) and O = K X (:73), on the 16-bit matrix pipe with error-compensated operands -- every
 * fp32 operand cut into 2 or 3 16-bit planes, the significant plane products accumulated in fp32 (csrc/meanshift_split.hip).
 * mode: PRIFIT_SPLIT_BF16X3 (2 bf16 planes, 3 products: operands to 2^-16), PRIFIT_SPLIT_BF16X6 (3 planes, 6 products:
 * 2^-24), PRIFIT_SPLIT_FP16X3 (2 fp16 planes of power-of-two scaled operands, 3 products: ~2^-22).  D == 128, N % 256 == 0.
 * _prep cuts the dictionary X [B,N,128] once per mean-shift call into `workspace` (_workspace BYTES, 16-byte aligned);
 * _fwd: one update's O [B,N,128] and rowsum [B,N] from the current points Z [B,N,128] (exponent, clamp and row sums in fp32
 * as in prifit_meanshift_fused_fwd); prifit_meanshift_update_fwd finishes the update. */
#define PRIFIT_SPLIT_BF16X3 1
#define PRIFIT_SPLIT_BF16X6 2
#define PRIFIT_SPLIT_FP16X3 3
int prifit_meanshift_split_supported(int N, int D, int mode);
long long prifit_meanshift_split_workspace(int B, int N, int D, int mode);
int prifit_meanshift_split_prep(const float *X, int B, int N, int D, int mode, void *workspace, void *stream);
int prifit_meanshift_split_fwd(const float *Z, const void *workspace, const float *bw, int B, int N, int D, int mode,
                               float *O, float *rowsum, void *stream);

/* Non-maximum suppression, src/mean_shift.py:162-202 called as nms(Z, Z, b) (:44).
 * dist [B,N,N] = 2 - 2 Z Z^T, Z [B,N,D], bw [B].  Outputs: owner [B,N] (nearest centre of each point),
 * counts [B,N], flags [B,N] (scratch), ids [B,cap] ascending kept centre ids, count [B] = number of kept
 * centres (may exceed cap: only the first cap ids are stored), labels [B,N] = argmax_k <Z[ids[k]], z_j>
 * over the stored centres, used [B,cap] = 1 where label k occurs.  owner_key (may be NULL): the keys
 * prifit_chord_sym_f32 left while it wrote `dist`; then the owner pass does not read the matrix again. */
int prifit_nms(const float *dist, const float *Z, const float *bw, int B, int N, int D, int cap,
               const unsigned long long *owner_key, int32_t *owner, int32_t *counts, int32_t *flags, int32_t *ids,
               int32_t *count, int32_t *labels, int32_t *used, void *stream);
/* prifit_nms on the mask and keys prifit_chord_sym_mask left (N % 128 == 0; counts 16-byte aligned): same outputs, bit for bit. */
int prifit_nms_mask(const uint32_t *mask, const float *Z, int B, int N, int D, int cap, const unsigned long long *owner_key,
                    int32_t *owner, int32_t *counts, int32_t *flags, int32_t *ids, int32_t *count, int32_t *labels, int32_t *used,
                    void *stream);

/* The same with centres that are not the points, src/mean_shift.py:162-202 as written: nms(centers, X, b) with
 * centers [B,N,D] and X [B,N,D] two tables of the same row count (upstream's `cluster_nbrs[uniques] * num_mem_cluster`
 * broadcast needs centers.shape[0] == X.shape[0]; e.g. the commented call nms(new_X, X, b) of :43).
 * dist_xc [B,N,N] = 2 - 2 X C^T (row j: the distances of point j to every centre), dist_cc [B,N,N] = 2 - 2 C C^T.
 * Outputs as prifit_nms; labels[j] = argmax_k <C[ids[k]], X[j]>. */
int prifit_nms_pair(const float *dist_xc, const float *dist_cc, const float *C, const float *X, const float *bw, int B,
                    int N, int D, int cap, int32_t *owner, int32_t *counts, int32_t *flags, int32_t *ids, int32_t *count,
                    int32_t *labels, int32_t *used, void *stream);

/* Soft membership, src/mean_shift.py:230-247.  dots [B,N,KM] = <x_j, centre_k> (raw), bw [B], gmax [B] =
 * max over live (k, j) of dots / bw^2 (detached): W [B,N,KM] = softmax-like weights, 0 for k >= count[b]. */
int prifit_membership_fwd(const float *dots, const float *bw, const float *gmax, const int32_t *count,
                          int B, int N, int KM, float *W, void *stream);
int prifit_membership_bwd(const float *gW, const float *W, const float *dots, const float *bw,
                          const float *gmax, const int32_t *count, int B, int N, int KM, float *gdots,
                          void *stream);

/* gmax [B] of prifit_membership_fwd in one launch: (max over points j and live clusters k < count[b] of dots[b][j][k]) /
 * bw[b]^2 -- `sim.max()` of src/mean_shift.py:237-242 after the division by b^2 (a positive scale commutes with the max).
 * KM % 4 == 0, dots 16-byte aligned; workspace: prifit_membership_gmax_workspace(B) floats. */
long long prifit_membership_gmax_workspace(int B);   /* floats of scratch */
int prifit_membership_gmax(const float *dots, const float *bw, const int32_t *count, int B, int N, int KM, float *gmax,
                           float *workspace, void *stream);

/* bw [B] = mean over the N rows of sqrt(max(kth, 1e-6)) (src/mean_shift.py:158-160), kth [B*N] from
 * prifit_kth_smallest_rows. */
int prifit_bandwidth_from_kth(const float *kth, int B, int N, float *bw, void *stream);

/* The cluster-count check of guard_mean_shift (src/ellipsoid_utils.py:19-27) for all shapes after prifit_nms:
 * nuniq [B] (may be NULL) = distinct labels per shape (count[b] itself when more than `cap` centres were kept),
 * bad [1] = 1 when some shape has nuniq > max_clusters (the quantile-doubling retry is due) or more than `slots` kept
 * centres, else 0. */
int prifit_cluster_verdict(const int32_t *count, const int32_t *used, int B, int cap, int max_clusters, int slots,
                           int32_t *nuniq, int32_t *bad, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* weighted ellipsoid fit and analytic-chamfer loss terms                                       */
/* (fixed capacity: KM <= 64 cluster slots per shape, live when k < count[b] and valid[b][k])   */
/* ------------------------------------------------------------------------------------------ */

/* Floats of saved state per (shape, cluster) written by the forward fit for its backward. */
int prifit_fit_state_floats(void);

/* src/ellipsoid_fitting.py:19-69 + principal_axis_ellipsoid(mode="slow") :119-141, one workgroup per
 * (shape, cluster).  points [B,N,3], W [B,N,KM]; rnd = the U[0,1) 3x3 matrices of :38, element (b,k)
 * at rnd + b*rnd_stride_b + k*rnd_stride_k (strides 0 share one matrix).  canonical_signs != 0 pins the
 * SVD column signs (largest component positive).  Outputs r [B,KM,3] semi-axes, V [B,KM,3,3] principal
 * axes (after the reflection fix :133-135), c [B,KM,3], valid [B,KM] (0 where S0/S2 > 1e5, :43). */
int prifit_ellipsoid_fit_fwd(const float *points, const float *W, const int32_t *count, const float *rnd,
                             long long rnd_stride_b, long long rnd_stride_k, int canonical_signs, int B,
                             int N, int KM, float *r, float *V, float *c, int32_t *valid, float *state,
                             void *stream);

/* (g_r, g_V, g_c) -> gW [B,N,KM]: through the extents, the reflection fix, CustomSVD.backward
 * (src/fitting_utils.py:67-139), the covariance (+noise) and the weighted centre. */
int prifit_ellipsoid_fit_bwd(const float *points, const float *W, const int32_t *count,
                             const int32_t *valid, const float *rnd, long long rnd_stride_b,
                             long long rnd_stride_k, const float *state, const float *g_r,
                             const float *g_V, const float *g_c, int B, int N, int KM, float *gW,
                             void *stream);

/* convex_loss.py:313-328 + src/utils.py:410-411: for every target point the ellipsoid with the smallest
 * |sdf|; arg [B,M] its slot (-1: none), fval [B,M] the signed value, sum_sq [B] = sum of squares. */
int prifit_ellipsoid_sdf_fwd(const float *targets, int B, int M, const float *r, const float *V,
                             const float *c, const int32_t *valid, int KM, int32_t *arg, float *fval,
                             float *sum_sq, void *stream);
/* g_{r,V,c} += gscale[b] * d(sum_sq[b]) / d{r,V,c}  (outputs initialised by the caller). */
int prifit_ellipsoid_sdf_bwd(const float *targets, int B, int M, const float *r, const float *V,
                             const float *c, const int32_t *arg, const float *gscale, int KM, float *g_r,
                             float *g_V, float *g_c, void *stream);

/* Every live ellipsoid's SDF at every point (convex_loss.py:331-343 compute_sdf_ellipsoids_batch), for the
 * optional intersection term (:374-413): sdf [B,M,KM], 0 in dead slots; and its autograd
 * (g_{r,V,c} += d<g_sdf, sdf>/d{r,V,c}, outputs initialised by the caller). */
int prifit_ellipsoid_sdf_matrix_fwd(const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, int KM, float *sdf,
                                    void *stream);
int prifit_ellipsoid_sdf_matrix_bwd(const float *points, int B, int M, const float *r, const float *V,
                                    const float *c, const int32_t *valid, const float *g_sdf, int KM,
                                    float *g_r, float *g_V, float *g_c, void *stream);

/* src/ellipsoid_utils.py:87-107: n [B,KM] surface samples per ellipsoid (round(10000*area/sum area),
 * <=0 -> 100), off [B,KM+1] exclusive prefix (off[KM] = total, clipped to cap). */
int prifit_sample_budget(const float *r, const int32_t *valid, int B, int KM, int cap, int32_t *n,
                         int32_t *off, void *stream);

/* The segmentation loss F.cross_entropy(pred, target) of models/pointnet2_part_seg_msg.py:137-144 (mean over the rows) and its
 * gradient, x [P, ld] rows of C <= 64 class scores, target [P] int64 with torch's default semantics: rows whose label is -100
 * (ignore_index) contribute nothing and are left out of the mean's denominator; any other label outside [0, C) is an error --
 * torch stops with a device-side assert, here the loss and that row's gradient become NaN (loud, no host read-back).
 * loss [2] = (mean over the kept rows of lse_r - x[r, t_r], number of kept rows); lse [P] is kept for the backward,
 * dx[r, c] = (exp(x[r, c] - lse_r) - [c == t_r]) g[0] / kept (kept = loss + 1).  workspace: prifit_cross_entropy_workspace()
 * floats.  Per-workgroup partial sums in a fixed row order: the same bits from run to run. */
int prifit_cross_entropy_workspace(void);
int prifit_cross_entropy_fwd(const float *x, long long ld, const long long *target, long long P, int C, float *lse, float *workspace,
                             float *loss, void *stream);
int prifit_cross_entropy_bwd(const float *x, long long ld, const long long *target, const float *lse, const float *g,
                             const float *kept, long long P, int C, float *dx, long long ldd, void *stream);

/* The combination step of analytic_chamfer_distance (src/utils.py:417-426) in one launch: per shape
 * (d2_sum[b] / max(total[b], 1) + sdf_sum[b] / M) / 2, averaged over the shapes with at least one valid primitive
 * (valid [B,KM]); 0 when none has.  loss [1]; part [2][B] = the two per-shape halves; coef [2 B + 1] = what the backward
 * needs.  _bwd: g [1] = d L / d loss -> g_d2 [B], g_sdf [B]. */
int prifit_chamfer_combine_fwd(const float *d2_sum, const int32_t *total, const float *sdf_sum, const int32_t *valid, int B,
                               int KM, int M, float *loss, float *part, float *coef, void *stream);
int prifit_chamfer_combine_bwd(const float *g, const float *coef, int B, float *g_d2, float *g_sdf, void *stream);

/* Surface samples on the Fibonacci (U,V) table evaluated as src/sample_ellipsoid.py:55-63, their exact
 * nearest target (src/utils.py:413-416): nn_idx [B,cap], sum_d2 [B] = sum of squared distances.
 * workspace: prifit_sample_nn_workspace_floats(B, cap) floats of scratch (8-byte aligned): the search runs over
 * several target ranges in parallel and a second launch keeps the first minimum. */
long long prifit_sample_nn_workspace_floats(int B, int cap);
int prifit_sample_nn_fwd(const float *r, const float *V, const float *c, const int32_t *n,
                         const int32_t *off, int B, int KM, const float *targets, int M, int cap,
                         int32_t *nn_idx, float *sum_d2, float *workspace, void *stream);
int prifit_sample_nn_bwd(const float *r, const float *V, const float *c, const int32_t *n,
                         const int32_t *off, int B, int KM, const float *targets, int M, int cap,
                         const int32_t *nn_idx, const float *gscale, float *g_r, float *g_V, float *g_c,
                         void *stream);

/* The `--if_cuboid` variant of the same loss (convex_loss.py:72-76,89,99): the fitted (r, V, c) are read as
 * boxes with half-sides r.  Same arguments and outputs as the ellipsoid entry points above.
 *   sdf:    convex_loss.py:473-502  q = |V^T (p - c)| - r, sdf = ||relu(q)|| + min(max(q), 0)
 *   budget: src/ellipsoid_utils.py:186-193  area = 8 (ab + bc + ca), round(10000 * area / sum), <= 0 -> 100
 *   sample: src/sample_ellipsoid.py:65-96 on the build's deterministic box-surface parameters (face by
 *           cumulative area, R2 sequence inside the face) in place of trimesh's random even sampling. */
int prifit_cuboid_sdf_fwd(const float *targets, int B, int M, const float *r, const float *V, const float *c,
                          const int32_t *valid, int KM, int32_t *arg, float *fval, float *sum_sq,
                          void *stream);
int prifit_cuboid_sdf_bwd(const float *targets, int B, int M, const float *r, const float *V, const float *c,
                          const int32_t *arg, const float *gscale, int KM, float *g_r, float *g_V,
                          float *g_c, void *stream);
int prifit_cuboid_sdf_matrix_fwd(const float *points, int B, int M, const float *r, const float *V,
                                 const float *c, const int32_t *valid, int KM, float *sdf, void *stream);
int prifit_cuboid_sdf_matrix_bwd(const float *points, int B, int M, const float *r, const float *V,
                                 const float *c, const int32_t *valid, const float *g_sdf, int KM,
                                 float *g_r, float *g_V, float *g_c, void *stream);
int prifit_cuboid_sample_budget(const float *r, const int32_t *valid, int B, int KM, int cap, int32_t *n,
                                int32_t *off, void *stream);
int prifit_cuboid_sample_nn_fwd(const float *r, const float *V, const float *c, const int32_t *n,
                                const int32_t *off, int B, int KM, const float *targets, int M, int cap,
                                int32_t *nn_idx, float *sum_d2, float *workspace, void *stream);
int prifit_cuboid_sample_nn_bwd(const float *r, const float *V, const float *c, const int32_t *n,
                                const int32_t *off, int B, int KM, const float *targets, int M, int cap,
                                const int32_t *nn_idx, const float *gscale, float *g_r, float *g_V,
                                float *g_c, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* DGCNN graph ops (src/dgcnn.py, BASELINE.json configs[4])                                     */
/* ------------------------------------------------------------------------------------------ */

/* k nearest neighbours in feature space, src/dgcnn.py:9-27: G [B,N,N] = x^T x (raw inner products, from
 * prifit_gemm_f32), xx [B,N] = |x|^2; idx [B,N,k] = top-k of (-xx_i - (-2 G_ij)) - xx_j, descending,
 * ties to the lower index.  N <= 4096. */
int prifit_knn_topk(const float *G, const float *xx, int B, int N, int k, int32_t *idx, void *stream);
/* The same selection for the network's FIRST graph (src/dgcnn.py:171-175: knn on the xyz coordinates), straight from the
 * cloud x [B,N,3]: the values -|x_i|^2 + 2 <x_i, x_j> - |x_j|^2 are formed from an LDS copy of the cloud with the product
 * kernel's own fma chain and rounding (the same indices bit for bit as prifit_gemm_f32 + prifit_knn_topk), and the [B,N,N]
 * pairwise matrix is never written.  64 <= N <= 2048, k <= 64 (prifit_knn3_supported). */
int prifit_knn3_supported(int N, int k);
int prifit_knn3_topk(const float *x, int B, int N, int k, int32_t *idx, void *stream);

/* Edge features, src/dgcnn.py:74-107: out[(b,n,j), :] = [x[b,idx[b,n,j]] - x[b,n], x[b,n], 0-pad],
 * x [B,N,C] channels-last, out rows of ld_out >= 2C floats. */
int prifit_edge_gather(const float *x, const int32_t *idx, int B, int N, int C, int k, int ld_out,
                       float *out, void *stream);
/* Its autograd: dx [B,N,C] (initialised by the caller) += scatter of gout [B*N*k, ld_gout]. */
int prifit_edge_scatter(const float *gout, int ld_gout, const int32_t *idx, int B, int N, int C, int k,
                        float *dx, void *stream);

/* ---- gradient exchange: ONE sum all-reduce of the flat fp32 gradient bucket per optimizer
 * step over RCCL / xGMI, on the caller's stream -- what the gradient reduce of nn.DataParallel
 * (train_partseg_shapenet.py:248-250) becomes with one process per GPU.  RCCL is resolved at run time (the copy already
 * loaded in the process first).  Protocol: rank 0 calls prifit_comm_unique_id and hands the
 * prifit_comm_unique_id_bytes() bytes to every rank (any side channel); every rank calls prifit_comm_init with its
 * device current; then prifit_allreduce_flat(buf, count, comm, stream) in place; prifit_comm_destroy at the end. */
int prifit_comm_unique_id_bytes(void);
/* 1 when the entry points are bound to an RCCL that was ALREADY loaded in the process (e.g. by torch.distributed's nccl
 * backend), 0 when the library had to load the system copy itself (or found none): a caller that also uses another
 * RCCL client in the same process must see 1 -- two RCCL instances in one process are undefined. */
int prifit_comm_in_process(void);
int prifit_comm_unique_id(void *out);
int prifit_comm_init(void **comm, int nranks, int rank, const void *unique_id);
int prifit_allreduce_flat(float *buf, long long count, void *comm, void *stream);
int prifit_comm_destroy(void *comm);

/* Adam over the whole parameter set in one launch: train_partseg_shapenet.py:252-259 (torch.optim.Adam(lr, betas=(0.9, 0.999),
 * eps=1e-08, weight_decay)) and `optimizer.step()` at :398 / :451.  params / exp_avg / exp_avg_sq: three flat fp32 buffers of
 * `total` floats with one layout -- parameter s at [offsets[s], offsets[s] + lengths[s]), offsets multiples of
 * prifit_adam_flat_alignment() floats; grads [nparams] DEVICE array of device pointers, NULL = no gradient this step (the
 * parameter is skipped: no decay, no step count, as torch does); step_in / step_out [nparams] the per-parameter step counts before
 * / after (two different arrays: the caller alternates them); skip: optional device flag, non-zero = a no-op (discarded step).
 * Arithmetic of torch/optim/adam.py:_single_tensor_adam (L2 weight decay, lerp first moment, bias corrections in double). */
int prifit_adam_flat_alignment(void);
int prifit_adam_flat(float *params, float *exp_avg, float *exp_avg_sq, const float *const *grads, const int32_t *offsets,
                     const int32_t *lengths, int nparams, long long total, const int32_t *step_in, int32_t *step_out, float lr,
                     float beta1, float beta2, float eps, float weight_decay, const int32_t *skip, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PRIFIT_HIP_H */
