// PointNet++ index / grouping kernels for gfx950 (wave64).  Compiled with -ffp-contract=off; every
// distance uses explicit __f*_rn intrinsics so that the rounding sequence equals the reference's
// PyTorch-CPU arithmetic (see oracle/prifit_oracle.c for the scalar statement of the same recipes).
#include <algorithm>

#include "common.h"
#include "distance.h"

// ---------------------------------------------------------------------------------------------
// farthest point sampling: one 256-thread workgroup per shape, points in registers, cloud copy in
// LDS for the centroid broadcast, packed (distance, ~index) keys reduced by wave shuffles + LDS.
// ---------------------------------------------------------------------------------------------
template <int PPT>
__global__ __launch_bounds__(256) void fps_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                  const int64_t *__restrict__ start_idx,
                                                  int64_t *__restrict__ out_idx,
                                                  float *__restrict__ new_xyz)
{
    extern __shared__ float s_xyz[];  // [N*3] then 8 x u64 slots
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float *P = xyz + (size_t)b * N * 3;
    for (int i = tid; i < N * 3; i += 256) s_xyz[i] = P[i];
    unsigned long long *slots =
        reinterpret_cast<unsigned long long *>(s_xyz + ((N * 3 + 1) & ~1));
    __syncthreads();

    float px[PPT], py[PPT], pz[PPT], mind[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int n = tid + 256 * j;
        bool ok = n < N;
        px[j] = ok ? s_xyz[n * 3 + 0] : 0.f;
        py[j] = ok ? s_xyz[n * 3 + 1] : 0.f;
        pz[j] = ok ? s_xyz[n * 3 + 2] : 0.f;
        mind[j] = 1e10f;
    }
    int far = (int)start_idx[b];
    for (int it = 0; it < npoint; ++it) {
        const float cx = s_xyz[far * 3 + 0], cy = s_xyz[far * 3 + 1], cz = s_xyz[far * 3 + 2];
        if (tid == 0) {
            out_idx[(size_t)b * npoint + it] = far;
            if (new_xyz) {
                float *o = new_xyz + ((size_t)b * npoint + it) * 3;
                o[0] = cx; o[1] = cy; o[2] = cz;
            }
        }
        unsigned long long best = 0ull;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            int n = tid + 256 * j;
            float dx = __fsub_rn(px[j], cx), dy = __fsub_rn(py[j], cy), dz = __fsub_rn(pz[j], cz);
            float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            mind[j] = d < mind[j] ? d : mind[j];
            // distances are >= 0, so the IEEE bit pattern orders like the value; ~index makes the
            // lowest index win among equal distances (torch.max returns the first maximum).
            unsigned long long key = ((unsigned long long)__float_as_uint(mind[j]) << 32) |
                                     (unsigned)(0xffffffffu - (unsigned)n);
            key = n < N ? key : 0ull;
            best = key > best ? key : best;
        }
        // wave arg-max of the packed key in two 32-bit DPP reductions: the largest distance, then the largest ~index
        // (= lowest index) among the lanes that hold it
        const unsigned bhi = (unsigned)(best >> 32), blo = (unsigned)(best & 0xffffffffu);
        const unsigned mhi = wave_max_u32_dpp(bhi);
        const unsigned mlo = wave_max_u32_dpp(bhi == mhi ? blo : 0u);
        unsigned long long *sl = slots + (it & 1) * 4;
        if ((tid & 63) == 0) sl[tid >> 6] = ((unsigned long long)mhi << 32) | mlo;
        __syncthreads();
        unsigned long long a = sl[0], c = sl[1], e = sl[2], g = sl[3];
        a = a > c ? a : c;
        e = e > g ? e : g;
        a = a > e ? a : e;
        far = (int)(0xffffffffu - (unsigned)(a & 0xffffffffu));
    }
}

// ---------------------------------------------------------------------------------------------
// ball query: one wave per query point, all radii in one pass, ordered ballot compaction
// (first nsample in-ball indices in ascending order, padded with the first: pointnet_util.py:100-106)
// ---------------------------------------------------------------------------------------------
struct BallArgs {
    float r2[4];
    int nsample[4];
    void *out[4];
};

constexpr int BQ_TILE = 2048;   // points staged in LDS per pass (32 KiB as float4)
constexpr int BQ_QPB = 4;       // queries per block: one per wave

// Two stages per wave.  Stage 1 scans the cloud 64 points at a time and only compacts the candidates of the LARGEST
// ball (index, distance) into a 128-entry per-wave ring, in index order: one compare + one ballot per chunk, no
// per-radius work, so successive chunks are independent.  Stage 2 runs whenever 64 candidates are pending (and once
// at the end) and does the ordered per-radius compaction on those 64 only: a few % of the points on real clouds.
template <int R, typename IdxT>
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ xyz,
                                                         const float *__restrict__ new_xyz, int N,
                                                         int S, BallArgs args)
{
    __shared__ float4 s_pts[BQ_TILE];
    __shared__ float2 s_cand[4][128];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *P = xyz + (size_t)b * N * 3;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    const int qid = blockIdx.x * BQ_QPB + wave;
    const bool valid = qid < S;   // wave-uniform
    const float *Q = new_xyz + ((size_t)b * S + (valid ? qid : S - 1)) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    const float qq = norm2_3(qx, qy, qz);
    int cnt[R], first[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { cnt[r] = 0; first[r] = N; }
    float r2max = args.r2[0];
#pragma unroll
    for (int r = 1; r < R; ++r) r2max = fmaxf(r2max, args.r2[r]);

    int fill = 0, done = 0;       // candidates written / consumed (wave-uniform)
    bool finished = !valid;       // every radius has its nsample indices
    float2 *ring = s_cand[wave];

    auto consume = [&](int nproc) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool act = lane < nproc;
        const float2 e = ring[(done + lane) & 127];
        const int gi = __float_as_int(e.x);
        const float d = e.y;
        bool all = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int K = args.nsample[r];
            if (cnt[r] < K) {
                const bool pred = act && !(d > args.r2[r]);
                const unsigned long long m = __ballot(pred);
                if (m != 0ull) {
                    if (cnt[r] == 0) first[r] = __builtin_amdgcn_readlane(gi, __builtin_ctzll(m));
                    const int pos = cnt[r] + __popcll(m & lt_mask);
                    if (pred && pos < K)
                        reinterpret_cast<IdxT *>(args.out[r])[((size_t)b * S + qid) * K + pos] = (IdxT)gi;
                    cnt[r] += __popcll(m);
                }
            }
            all = all && cnt[r] >= K;
        }
        done += nproc;
        __builtin_amdgcn_wave_barrier();
        return all;
    };

    for (int base = 0; base < N; base += BQ_TILE) {
        const int tn = min(BQ_TILE, N - base);
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += 256) {
            float x = P[(size_t)(base + i) * 3 + 0], y = P[(size_t)(base + i) * 3 + 1],
                  z = P[(size_t)(base + i) * 3 + 2];
            s_pts[i] = make_float4(x, y, z, norm2_3(x, y, z));
        }
        __syncthreads();
        if (finished) continue;
        for (int c = 0; c < tn; c += 64) {
            const int i = c + lane;
            const bool inb = i < tn;
            const float4 p = s_pts[inb ? i : 0];
            const float d = sqdist_expanded(qx, qy, qz, qq, p.x, p.y, p.z, p.w);
            const bool pred = inb && !(d > r2max);
            const unsigned long long m = __ballot(pred);
            if (m != 0ull) {
                if (pred) ring[(fill + __popcll(m & lt_mask)) & 127] = make_float2(__int_as_float(base + i), d);
                fill += __popcll(m);
                if (fill - done >= 64 && consume(64)) { finished = true; break; }
            }
        }
    }
    if (valid) {
        while (!finished && fill > done) finished = consume(min(64, fill - done));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int K = args.nsample[r];
            IdxT *o = reinterpret_cast<IdxT *>(args.out[r]) + ((size_t)b * S + qid) * K;
            for (int k = min(cnt[r], K) + lane; k < K; k += 64) o[k] = (IdxT)first[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// three nearest neighbours (thread per query, candidate cloud staged in LDS)
// ---------------------------------------------------------------------------------------------
constexpr int NN_TILE = 2048;

__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2, int N, int S,
                                                       int32_t *__restrict__ idx,
                                                       float *__restrict__ dist,
                                                       float *__restrict__ weight)
{
    __shared__ float4 s_pts[NN_TILE];
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool ok = n < N;
    const float *Q = xyz1 + ((size_t)b * N + (ok ? n : 0)) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    const float qq = norm2_3(qx, qy, qz);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    const float *P = xyz2 + (size_t)b * S * 3;
    for (int base = 0; base < S; base += NN_TILE) {
        const int tn = min(NN_TILE, S - base);
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += 256) {
            float x = P[(size_t)(base + i) * 3], y = P[(size_t)(base + i) * 3 + 1],
                  z = P[(size_t)(base + i) * 3 + 2];
            s_pts[i] = make_float4(x, y, z, norm2_3(x, y, z));
        }
        __syncthreads();
        for (int i = 0; i < tn; ++i) {
            const float4 p = s_pts[i];
            const float d = sqdist_expanded(qx, qy, qz, qq, p.x, p.y, p.z, p.w);
            const int s = base + i;
            if (d < d2) {  // strict: equal distances keep the earlier (lower) index
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = s; }
                    else { d1 = d; i1 = s; }
                } else { d2 = d; i2 = s; }
            }
        }
    }
    if (!ok) return;
    const size_t o = ((size_t)b * N + n) * 3;
    idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
    if (dist) { dist[o] = d0; dist[o + 1] = d1; dist[o + 2] = d2; }
    // pointnet_util.py:295-297: recip = 1/(d + 1e-8); weight = recip / sum(recip)
    const float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f));
    const float r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f));
    const float r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
    const float nrm = __fadd_rn(__fadd_rn(r0, r1), r2);
    weight[o] = __fdiv_rn(r0, nrm);
    weight[o + 1] = __fdiv_rn(r1, nrm);
    weight[o + 2] = __fdiv_rn(r2, nrm);
}

__global__ __launch_bounds__(256) void square_distance_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ dst, int S,
                                                              int N, float *__restrict__ out)
{
    const int b = blockIdx.z, s = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float *Q = src + ((size_t)b * S + s) * 3;
    const float *P = dst + ((size_t)b * N + n) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2], px = P[0], py = P[1], pz = P[2];
    out[((size_t)b * S + s) * N + n] =
        sqdist_expanded(qx, qy, qz, norm2_3(qx, qy, qz), px, py, pz, norm2_3(px, py, pz));
}

// ---------------------------------------------------------------------------------------------
// grouping gather / scatter-add, 3-NN interpolation (bandwidth kernels)
// ---------------------------------------------------------------------------------------------
// Vector path: C % 4 == 0, order 0 ([feat, rel, pad]), ld_out % 4 == 0.  One float4 per lane, consecutive
// lanes write consecutive float4s of the output (fully coalesced 1 KiB per wave-instruction); every thread
// keeps GG_UNROLL independent (index -> row -> store) chains in flight to cover the dependent-load latency.
constexpr int GG_UNROLL = 4;
__global__ __launch_bounds__(256) void group_gather_vec_kernel(
    const float4 *__restrict__ feat, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int32_t *__restrict__ idx, int N, int S, int K, int C4, int V, long long total_vec,
    float4 *__restrict__ out)
{
    // 32-bit index arithmetic (the launcher guarantees total_vec < 2^31): 64-bit divisions are emulated
    const unsigned total = (unsigned)total_vec;
    for (unsigned base = blockIdx.x * (256u * GG_UNROLL); base < total; base += gridDim.x * (256u * GG_UNROLL)) {
        unsigned id[GG_UNROLL], bs[GG_UNROLL];
        int v[GG_UNROLL], n[GG_UNROLL];
        bool ok[GG_UNROLL];
#pragma unroll
        for (int u = 0; u < GG_UNROLL; ++u) {
            id[u] = base + u * 256u + threadIdx.x;
            ok[u] = id[u] < total;
            const unsigned row = (ok[u] ? id[u] : 0u) / (unsigned)V;
            v[u] = (int)((ok[u] ? id[u] : 0u) - row * (unsigned)V);
            bs[u] = row / (unsigned)K;  // b*S + s
            n[u] = idx[row];
        }
        float4 val[GG_UNROLL];
#pragma unroll
        for (int u = 0; u < GG_UNROLL; ++u) {
            const int b = (int)(bs[u] / (unsigned)S);
            const bool in = n[u] >= 0 && n[u] < N;
            const int nn = in ? n[u] : 0;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v[u] < C4) {
                x = feat[((size_t)b * N + nn) * C4 + v[u]];
            } else if (v[u] == C4) {
                const float *p = xyz + ((size_t)b * N + nn) * 3;
                const float *c = new_xyz + (size_t)bs[u] * 3;
                x = make_float4(p[0] - c[0], p[1] - c[1], p[2] - c[2], 0.f);
            }
            val[u] = in ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < GG_UNROLL; ++u)
            if (ok[u]) out[id[u]] = val[u];
    }
}

// Narrow rows (SA1: C = 3 features + 3 relative coordinates in an 8-float row): one thread per row, two
// float4 stores, consecutive lanes on consecutive rows (2 KiB contiguous per wave).
__global__ __launch_bounds__(256) void group_gather_row8_kernel(
    const float *__restrict__ feat, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int32_t *__restrict__ idx, int N, int S, int K, int C, unsigned rows, float4 *__restrict__ out)
{
    for (unsigned row = blockIdx.x * 256u + threadIdx.x; row < rows; row += gridDim.x * 256u) {
        const unsigned bs = row / (unsigned)K;
        const int b = (int)(bs / (unsigned)S);
        const int n = idx[row];
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n >= 0 && n < N) {
            const float *f = feat + ((size_t)b * N + n) * C;
            for (int c = 0; c < C; ++c) v[c] = f[c];
            const float *p = xyz + ((size_t)b * N + n) * 3;
            const float *ctr = new_xyz + (size_t)bs * 3;
            v[C] = p[0] - ctr[0]; v[C + 1] = p[1] - ctr[1]; v[C + 2] = p[2] - ctr[2];
        }
        out[(size_t)row * 2] = make_float4(v[0], v[1], v[2], v[3]);
        out[(size_t)row * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// Scalar path: any C / order / ld_out.
__global__ __launch_bounds__(256) void group_gather_scalar_kernel(
    const float *__restrict__ feat, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int32_t *__restrict__ idx, int N, int S, int K, int C, int order, int ld, long long total,
    float *__restrict__ out)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total;
         id += (long long)gridDim.x * 256) {
        const long long row = id / ld;
        const int c = (int)(id - row * ld);
        const long long bs = row / K;
        const int b = (int)(bs / S);
        const int n = idx[row];
        float val = 0.f;
        if (n >= 0 && n < N) {
            int fc, rc;  // feature column / rel column, -1 if this output column is neither
            if (order == 0) { fc = c < C ? c : -1; rc = (c >= C && c < C + 3) ? c - C : -1; }
            else { rc = c < 3 ? c : -1; fc = (c >= 3 && c < C + 3) ? c - 3 : -1; }
            if (fc >= 0) val = feat[((size_t)b * N + n) * C + fc];
            else if (rc >= 0) val = xyz[((size_t)b * N + n) * 3 + rc] - new_xyz[(size_t)bs * 3 + rc];
        }
        out[id] = val;
    }
}

// dfeat[b, idx, c] += gout[row, col0 + c]; one lane per float so that each wave-instruction adds
// 256 contiguous bytes (the shape that runs at the chip-wide float-atomic rate).
__global__ __launch_bounds__(256) void group_scatter_add_kernel(
    const float *__restrict__ gout, int ld, int col0, const int32_t *__restrict__ idx, int N, int S,
    int K, int C, long long total, float *__restrict__ dfeat)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total;
         id += (long long)gridDim.x * 256) {
        const long long row = id / C;
        const int c = (int)(id - row * C);
        const int b = (int)(row / ((long long)S * K));
        const int n = idx[row];
        if (n >= 0 && n < N) unsafeAtomicAdd(dfeat + ((size_t)b * N + n) * C + c, gout[row * ld + col0 + c]);
    }
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(
    const float *__restrict__ points2, const int32_t *__restrict__ idx, const float *__restrict__ w,
    int N, int S, int C, int ld, int col0, long long total, float *__restrict__ out)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total;
         id += (long long)gridDim.x * 256) {
        const long long row = id / C;  // b*N + n
        const int c = (int)(id - row * C);
        const int b = (int)(row / N);
        const int32_t *ii = idx + row * 3;
        const float *ww = w + row * 3;
        const float *P = points2 + (size_t)b * S * C;
        // (p0*w0 + p1*w1) + p2*w2, the order of torch.sum over the 3-slot axis (pointnet_util.py:298)
        float acc = P[(size_t)ii[0] * C + c] * ww[0];
        acc += P[(size_t)ii[1] * C + c] * ww[1];
        acc += P[(size_t)ii[2] * C + c] * ww[2];
        out[row * ld + col0 + c] = acc;
    }
}

// The whole input row of a feature-propagation MLP in one pass (models/pointnet_util.py:296-306): [interpolated | points1 | 0-pad]
// (the build's internal order; upstream concatenates [points1, interpolated]) -- the interpolation, the concatenation and the
// padding were a kernel + a cat (+ an expand copy when S == 1).  idx == NULL: S == 1, every point takes points2[b, 0] (:287-288).
__global__ __launch_bounds__(256) void fp_rows_kernel(const float *__restrict__ points2, const int32_t *__restrict__ idx,
                                                      const float *__restrict__ w, const float *__restrict__ points1, int N, int S,
                                                      int D2, int D1, int kp, long long total, float *__restrict__ out)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long row = id / kp;  // b*N + n
        const int c = (int)(id - row * kp);
        float v = 0.f;
        if (c < D2) {
            const float *P = points2 + (size_t)(row / N) * S * D2;
            if (idx) {
                const int32_t *ii = idx + row * 3;
                const float *ww = w + row * 3;
                // (p0*w0 + p1*w1) + p2*w2, the order of torch.sum over the 3-slot axis (pointnet_util.py:298)
                v = P[(size_t)ii[0] * D2 + c] * ww[0];
                v += P[(size_t)ii[1] * D2 + c] * ww[1];
                v += P[(size_t)ii[2] * D2 + c] * ww[2];
            } else {
                v = P[c];
            }
        } else if (c < D2 + D1) {
            v = points1[row * D1 + (c - D2)];
        }
        out[id] = v;
    }
}

__global__ __launch_bounds__(256) void three_interpolate_bwd_kernel(
    const float *__restrict__ gout, int ld, int col0, const int32_t *__restrict__ idx,
    const float *__restrict__ w, int N, int S, int C, long long total, float *__restrict__ dp2)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total;
         id += (long long)gridDim.x * 256) {
        const long long row = id / C;
        const int c = (int)(id - row * C);
        const int b = (int)(row / N);
        const float g = gout[row * ld + col0 + c];
        float *D = dp2 + (size_t)b * S * C;
#pragma unroll
        for (int j = 0; j < 3; ++j) unsafeAtomicAdd(D + (size_t)idx[row * 3 + j] * C + c, g * w[row * 3 + j]);
    }
}

static inline int grid_for(long long total, int per_block = 256, int cap = 256 * 16)
{
    long long g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
// Column permutation / padding of a weight matrix and its autograd (the internal row layouts of the first layers differ from
// upstream's column order and are padded to 16-byte rows; models/pointnet_util.py:195-197, :306 decide the order).  One
// launch each way in place of index_select + mul (forward) and zeros + index_select + index_add_ (backward) per weight.
__global__ __launch_bounds__(256) void pack_cols_kernel(const float *__restrict__ w, int rows, int scols, const int32_t *__restrict__ map,
                                                        int dcols, float *__restrict__ out)
{
    const int total = rows * dcols;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int r = id / dcols, j = id - r * dcols;
        const int c = map[j];
        out[id] = c >= 0 ? w[(size_t)r * scols + c] : 0.f;
    }
}

// gw[r][c] = sum over the output columns j fed by source column c (inv_idx[inv_off[c] .. inv_off[c+1]), ascending) of g[r][j]
__global__ __launch_bounds__(256) void unpack_cols_kernel(const float *__restrict__ g, int rows, int dcols,
                                                          const int32_t *__restrict__ inv_off, const int32_t *__restrict__ inv_idx,
                                                          int scols, float *__restrict__ gw)
{
    const int total = rows * scols;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int r = id / scols, c = id - r * scols;
        float acc = 0.f;
        for (int e = inv_off[c]; e < inv_off[c + 1]; ++e) acc += g[(size_t)r * dcols + inv_idx[e]];
        gw[id] = acc;
    }
}

// Every packed weight of a network in ONE launch per direction (blockIdx.y = the job); the jobs travel by value.
constexpr int PACK_JOBS = 16;
struct PackJobs {
    const float *src[PACK_JOBS];
    float *dst[PACK_JOBS];
    const int32_t *map[PACK_JOBS];       // pack: map; unpack: inv_off
    const int32_t *idx[PACK_JOBS];       // unpack: inv_idx
    int rows[PACK_JOBS], scols[PACK_JOBS], dcols[PACK_JOBS];
};

__global__ __launch_bounds__(256) void pack_cols_multi_kernel(PackJobs jb)
{
    const int j = blockIdx.y;
    const float *__restrict__ w = jb.src[j];
    float *__restrict__ out = jb.dst[j];
    const int32_t *__restrict__ map = jb.map[j];
    const int scols = jb.scols[j], dcols = jb.dcols[j], total = jb.rows[j] * dcols;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int r = id / dcols, c = map[id - r * dcols];
        out[id] = c >= 0 ? w[(size_t)r * scols + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void unpack_cols_multi_kernel(PackJobs jb)
{
    const int j = blockIdx.y;
    const float *__restrict__ g = jb.src[j];
    float *__restrict__ gw = jb.dst[j];
    const int32_t *__restrict__ inv_off = jb.map[j], *__restrict__ inv_idx = jb.idx[j];
    const int scols = jb.scols[j], dcols = jb.dcols[j], total = jb.rows[j] * scols;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int r = id / scols, c = id - r * scols;
        float acc = 0.f;
        for (int e = inv_off[c]; e < inv_off[c + 1]; ++e) acc += g[(size_t)r * dcols + inv_idx[e]];
        gw[id] = acc;
    }
}

extern "C" {

int prifit_version(const char **arch)
{
    if (arch) *arch = "gfx950";
    return 100;
}

int prifit_fps(const float *xyz, int B, int N, int npoint, const int64_t *start_idx, int64_t *out_idx,
               float *new_xyz, void *stream)
{
    if (!xyz || !start_idx || !out_idx || B <= 0 || N <= 0 || npoint <= 0 || N > 4096)
        return PRIFIT_EINVAL;
    const size_t lds = (size_t)((N * 3 + 1) & ~1) * sizeof(float) + 8 * sizeof(unsigned long long);
    hipStream_t st = as_stream(stream);
    const int ppt = (N + 255) / 256;
#define FPS_LAUNCH(P) \
    hipLaunchKernelGGL((fps_kernel<P>), dim3(B), dim3(256), lds, st, xyz, N, npoint, start_idx, out_idx, new_xyz)
    if (ppt <= 1) FPS_LAUNCH(1);
    else if (ppt <= 2) FPS_LAUNCH(2);
    else if (ppt <= 4) FPS_LAUNCH(4);
    else if (ppt <= 8) FPS_LAUNCH(8);
    else FPS_LAUNCH(16);
#undef FPS_LAUNCH
    return prifit_check_launch();
}

int prifit_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, int R,
                      const float *radius2, const int *nsample, void *const *out, int idx64, void *stream)
{
    if (!xyz || !new_xyz || !radius2 || !nsample || !out || B <= 0 || N <= 0 || S <= 0 || R < 1 || R > 4)
        return PRIFIT_EINVAL;
    BallArgs a;
    for (int r = 0; r < 4; ++r) {
        a.r2[r] = r < R ? radius2[r] : 0.f;
        a.nsample[r] = r < R ? nsample[r] : 0;
        a.out[r] = r < R ? out[r] : nullptr;
        if (r < R && (nsample[r] < 1 || nsample[r] > 1024 || !out[r])) return PRIFIT_EINVAL;
    }
    dim3 grid((S + BQ_QPB - 1) / BQ_QPB, B), block(256);
    hipStream_t st = as_stream(stream);
#define BQ_LAUNCH(RR, T) hipLaunchKernelGGL((ball_query_kernel<RR, T>), grid, block, 0, st, xyz, new_xyz, N, S, a)
#define BQ_DISPATCH(T)                  \
    switch (R) {                        \
        case 1: BQ_LAUNCH(1, T); break; \
        case 2: BQ_LAUNCH(2, T); break; \
        case 3: BQ_LAUNCH(3, T); break; \
        default: BQ_LAUNCH(4, T); break; \
    }
    if (idx64) { BQ_DISPATCH(int64_t) } else { BQ_DISPATCH(int32_t) }
#undef BQ_DISPATCH
#undef BQ_LAUNCH
    return prifit_check_launch();
}

int prifit_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int32_t *idx, float *dist,
                    float *weight, void *stream)
{
    if (!xyz1 || !xyz2 || !idx || !weight || B <= 0 || N <= 0 || S < 3) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(three_nn_kernel, dim3((N + 255) / 256, B), dim3(256), 0, as_stream(stream), xyz1,
                       xyz2, N, S, idx, dist, weight);
    return prifit_check_launch();
}

int prifit_square_distance(const float *src, const float *dst, int B, int S, int N, float *out, void *stream)
{
    if (!src || !dst || !out || B <= 0 || S <= 0 || N <= 0 || S > 65535 || B > 65535) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(square_distance_kernel, dim3((N + 255) / 256, S, B), dim3(256), 0, as_stream(stream),
                       src, dst, S, N, out);
    return prifit_check_launch();
}

int prifit_group_gather(const float *feat, const float *xyz, const float *new_xyz, const int32_t *idx, int B,
                        int N, int S, int K, int C, int order, int ld_out, float *out, void *stream)
{
    if (!xyz || !new_xyz || !idx || !out || (C > 0 && !feat) || B <= 0 || N <= 0 || S <= 0 || K <= 0 ||
        C < 0 || ld_out < C + 3 || (order != 0 && order != 1))
        return PRIFIT_EINVAL;
    const long long rows = (long long)B * S * K;
    hipStream_t st = as_stream(stream);
    const bool vec = order == 0 && C > 0 && (C % 4 == 0) && (ld_out % 4 == 0) &&
                     ((uintptr_t)feat % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                     rows * (ld_out / 4) < 2147483647LL;
    if (vec) {
        const int V = ld_out / 4;
        const long long total = rows * V;
        hipLaunchKernelGGL(group_gather_vec_kernel, dim3(grid_for(total, 256 * GG_UNROLL, 256 * 16)), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(feat), xyz, new_xyz, idx, N, S, K, C / 4, V,
                           total, reinterpret_cast<float4 *>(out));
    } else if (order == 0 && C <= 5 && ld_out == 8 && ((uintptr_t)out % 16 == 0) && rows < 2147483647LL) {
        hipLaunchKernelGGL(group_gather_row8_kernel, dim3(grid_for(rows, 256, 256 * 16)), dim3(256), 0, st, feat, xyz,
                           new_xyz, idx, N, S, K, C, (unsigned)rows, reinterpret_cast<float4 *>(out));
    } else {
        const long long total = rows * ld_out;
        hipLaunchKernelGGL(group_gather_scalar_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, st,
                           feat, xyz, new_xyz, idx, N, S, K, C, order, ld_out, total, out);
    }
    return prifit_check_launch();
}

int prifit_group_scatter_add(const float *gout, int ld_gout, int col0, const int32_t *idx, int B, int N, int S,
                             int K, int C, float *dfeat, void *stream)
{
    if (!gout || !idx || !dfeat || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0 || col0 < 0 ||
        ld_gout < col0 + C)
        return PRIFIT_EINVAL;
    const long long total = (long long)B * S * K * C;
    hipLaunchKernelGGL(group_scatter_add_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0,
                       as_stream(stream), gout, ld_gout, col0, idx, N, S, K, C, total, dfeat);
    return prifit_check_launch();
}

int prifit_three_interpolate(const float *points2, const int32_t *idx, const float *weight, int B, int N, int S,
                             int C, int ld_out, int col0, float *out, void *stream)
{
    if (!points2 || !idx || !weight || !out || B <= 0 || N <= 0 || S <= 0 || C <= 0 || col0 < 0 ||
        ld_out < col0 + C)
        return PRIFIT_EINVAL;
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0,
                       as_stream(stream), points2, idx, weight, N, S, C, ld_out, col0, total, out);
    return prifit_check_launch();
}

int prifit_fp_rows(const float *points2, const int32_t *idx, const float *weight, const float *points1, int B, int N, int S, int D2,
                   int D1, int kp, float *out, void *stream)
{
    if (!points2 || !out || B <= 0 || N <= 0 || S <= 0 || D2 <= 0 || D1 < 0 || kp < D2 + D1 || (D1 > 0 && !points1) ||
        (idx ? !weight : S != 1))
        return PRIFIT_EINVAL;
    const long long total = (long long)B * N * kp;
    hipLaunchKernelGGL(fp_rows_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, as_stream(stream), points2, idx, weight,
                       points1, N, S, D2, D1, kp, total, out);
    return prifit_check_launch();
}

int prifit_three_interpolate_bwd(const float *gout, int ld_gout, int col0, const int32_t *idx,
                                 const float *weight, int B, int N, int S, int C, float *dpoints2, void *stream)
{
    if (!gout || !idx || !weight || !dpoints2 || B <= 0 || N <= 0 || S <= 0 || C <= 0 || col0 < 0 ||
        ld_gout < col0 + C)
        return PRIFIT_EINVAL;
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(three_interpolate_bwd_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0,
                       as_stream(stream), gout, ld_gout, col0, idx, weight, N, S, C, total, dpoints2);
    return prifit_check_launch();
}

int prifit_pack_cols(const float *w, int rows, int src_cols, const int32_t *map, int dst_cols, float *out, void *stream)
{
    if (!w || !map || !out || rows <= 0 || src_cols <= 0 || dst_cols <= 0 || (long long)rows * dst_cols > 0x7fffffffLL)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(pack_cols_kernel, dim3(grid_for((long long)rows * dst_cols, 256, 256 * 8)), dim3(256), 0, as_stream(stream),
                       w, rows, src_cols, map, dst_cols, out);
    return prifit_check_launch();
}

int prifit_unpack_cols(const float *g, int rows, int dst_cols, const int32_t *inv_off, const int32_t *inv_idx, int src_cols,
                       float *gw, void *stream)
{
    if (!g || !inv_off || !inv_idx || !gw || rows <= 0 || src_cols <= 0 || dst_cols <= 0 || (long long)rows * src_cols > 0x7fffffffLL)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(unpack_cols_kernel, dim3(grid_for((long long)rows * src_cols, 256, 256 * 8)), dim3(256), 0,
                       as_stream(stream), g, rows, dst_cols, inv_off, inv_idx, src_cols, gw);
    return prifit_check_launch();
}

int prifit_pack_cols_max_jobs(void) { return PACK_JOBS; }

int prifit_pack_cols_multi(int njobs, const float *const *w, const int32_t *rows, const int32_t *src_cols, const int32_t *const *map,
                           const int32_t *dst_cols, float *const *out, void *stream)
{
    if (njobs <= 0 || njobs > PACK_JOBS || !w || !rows || !src_cols || !map || !dst_cols || !out) return PRIFIT_EINVAL;
    PackJobs jb = {};
    long long most = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!w[j] || !map[j] || !out[j] || rows[j] <= 0 || src_cols[j] <= 0 || dst_cols[j] <= 0 || (long long)rows[j] * dst_cols[j] > 0x7fffffffLL)
            return PRIFIT_EINVAL;
        jb.src[j] = w[j]; jb.dst[j] = out[j]; jb.map[j] = map[j]; jb.rows[j] = rows[j]; jb.scols[j] = src_cols[j]; jb.dcols[j] = dst_cols[j];
        most = std::max(most, (long long)rows[j] * dst_cols[j]);
    }
    hipLaunchKernelGGL(pack_cols_multi_kernel, dim3(grid_for(most, 256, 256 * 8), njobs), dim3(256), 0, as_stream(stream), jb);
    return prifit_check_launch();
}

int prifit_unpack_cols_multi(int njobs, const float *const *g, const int32_t *rows, const int32_t *dst_cols, const int32_t *const *inv_off,
                             const int32_t *const *inv_idx, const int32_t *src_cols, float *const *gw, void *stream)
{
    if (njobs <= 0 || njobs > PACK_JOBS || !g || !rows || !dst_cols || !inv_off || !inv_idx || !src_cols || !gw) return PRIFIT_EINVAL;
    PackJobs jb = {};
    long long most = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!g[j] || !inv_off[j] || !inv_idx[j] || !gw[j] || rows[j] <= 0 || src_cols[j] <= 0 || dst_cols[j] <= 0 ||
            (long long)rows[j] * src_cols[j] > 0x7fffffffLL)
            return PRIFIT_EINVAL;
        jb.src[j] = g[j]; jb.dst[j] = gw[j]; jb.map[j] = inv_off[j]; jb.idx[j] = inv_idx[j]; jb.rows[j] = rows[j];
        jb.scols[j] = src_cols[j]; jb.dcols[j] = dst_cols[j];
        most = std::max(most, (long long)rows[j] * src_cols[j]);
    }
    hipLaunchKernelGGL(unpack_cols_multi_kernel, dim3(grid_for(most, 256, 256 * 8), njobs), dim3(256), 0, as_stream(stream), jb);
    return prifit_check_launch();
}

}  // extern "C"
