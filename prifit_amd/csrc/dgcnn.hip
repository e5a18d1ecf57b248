// DGCNN graph ops (src/dgcnn.py): k-nearest-neighbour selection on a pairwise matrix and the edge-feature
// gather / scatter of get_graph_feature.  The pairwise inner products come from the MFMA GEMM; the
// edge convolutions and their GroupNorm run on the GEMM + normalisation kernels of gemm.hip / bn.hip.
#include "common.h"

// idx[row][0..k) = indices of the k largest v_j = (-xx_i - (-2 G_ij)) - xx_j, descending, ties to the lower
// index (src/dgcnn.py:15-22: pairwise_distance.topk(k)).  One wave per row, the row lives in registers.
template <int VPT>
__global__ __launch_bounds__(256) void knn_topk_kernel(const float *__restrict__ G, const float *__restrict__ xx,
                                                       int N, long long rows, int k, int32_t *__restrict__ idx)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const long long b = row / N;
    const float *g = G + row * N;
    const float *xb = xx + b * N;
    const float nxi = -xb[row - b * N];
    // branch-free loads (clamped column), padding afterwards: predicated loads are serialised by the compiler
    float v[VPT], gv[VPT], xv[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        gv[j] = g[c < N ? c : N - 1];
        xv[j] = xb[c < N ? c : N - 1];
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) v[j] = (lane + 64 * j) < N ? (nxi - (-2.0f * gv[j])) - xv[j] : -INFINITY;
    for (int t = 0; t < k; ++t) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < VPT; ++j)
            if (v[j] > best) { best = v[j]; bi = lane + 64 * j; }  // ascending index within the lane: first max
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) idx[row * k + t] = bi;
#pragma unroll
        for (int j = 0; j < VPT; ++j)
            if (lane + 64 * j == bi) v[j] = -INFINITY;
    }
}

// The same selection without k rounds of (scan 32 values per lane, wave arg-max, invalidate): (1) the k-th largest value by
// bisection on the order-preserving key -- one vector compare per key and round, the count is the scalar popcount of the
// lane mask; (2) everything above it plus the lowest-index ties are compacted into LDS through ballot prefixes;
// (3) one bitonic sort of <= 64 (key, ~index) pairs per wave puts them in topk's order (descending value, ties to the
// lower index).  Same indices bit for bit; 371 -> 306 us at N = 2048, k = 20 (staging the squared norms through LDS: no gain).  k <= 64.
__device__ __forceinline__ unsigned knn_key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Selection of the k largest of a row's keys (order-preserving images of the values, 0 = padding), VPT per lane, column of
// key[j] = lane + 64 j.  out[0..k) = their columns, descending value, ties to the lower column.  `sel`: 64 wave-private LDS slots.
template <int VPT>
__device__ __forceinline__ void knn_select_keys(unsigned (&key)[VPT], int N, int k, unsigned long long *sel, int lane,
                                                int32_t *__restrict__ out)
{
    // (0) fast path (round 4).  The k-th largest of the 64 per-lane maxima, t0, is a conservative threshold: at least k keys
    // (those maxima) are >= t0, so the k largest keys of the row are all >= t0 -- and usually little more than k keys are.
    // When at most 64 keys pass, they are compacted and sorted right away: the composite (key, ~index) order of the sort IS
    // topk's order (descending value, ties to the lower index), so the result is the one the bisection below would give.
    // One compare + ballot per key instead of one per key and BISECTION ROUND (~12 rounds): 293 -> see DESIGN 5g.  Rows with
    // more than 64 keys >= t0 (clouds full of exact ties) take the general path.
    {
        unsigned lmax = 0u;
#pragma unroll
        for (int j = 0; j < VPT; ++j) lmax = max(lmax, key[j]);
        // t0 = the largest p with count(lane maxima >= p) >= k   (k <= 64 on this path)
        unsigned t0 = 0u;
        {
            const unsigned hi = wave_max_u32_dpp(lmax), lo = ~wave_max_u32_dpp(~lmax);
            t0 = lo;
            if (hi != lo) {
                const int top = 31 - __builtin_clz(hi ^ lo);
                t0 = top == 31 ? 0u : (lo >> (top + 1)) << (top + 1);
                for (int bit = top; bit >= 0; --bit) {
                    const unsigned p = t0 | (1u << bit);
                    if (__builtin_popcountll(__ballot(lmax >= p)) >= k) t0 = p;
                }
            }
        }
        const unsigned long long ltm = (1ull << lane) - 1ull;
        int n0 = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const bool take = key[j] >= t0 && key[j] != 0u;
            const unsigned long long mt = __ballot(take);
            const int pos = n0 + __builtin_popcountll(mt & ltm);
            if (take && pos < 64)
                sel[pos] = ((unsigned long long)key[j] << 32) | (unsigned)(~(unsigned)(lane + 64 * j));
            n0 += __builtin_popcountll(mt);
        }
        if (n0 >= k && n0 <= 64) {   // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned long long v = lane < n0 ? sel[lane] : 0ull;
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
                for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                    const unsigned lo2 = __shfl_xor((unsigned)v, stride, 64), hi2 = __shfl_xor((unsigned)(v >> 32), stride, 64);
                    const unsigned long long o = ((unsigned long long)hi2 << 32) | lo2;
                    const bool up = (lane & size) != 0;
                    const bool lower = (lane & stride) == 0;
                    const bool keep_max = up ? !lower : lower;
                    v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
                }
            }
            if (lane < k) out[lane] = (int)(~(unsigned)v);
            return;
        }
        __builtin_amdgcn_wave_barrier();   // (the general path below rewrites s_sel)
    }
    // (1) tau = the k-th largest key: the largest p with count(key >= p) >= k
    unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        kmax = max(kmax, key[j]);
        kmin = min(kmin, (lane + 64 * j) < N ? key[j] : 0xffffffffu);
    }
    kmax = wave_max_u32_dpp(kmax);
    kmin = ~wave_max_u32_dpp(~kmin);
    unsigned tau = kmin;
    if (kmin != kmax) {
        const int top = 31 - __builtin_clz(kmin ^ kmax);
        tau = top == 31 ? 0u : (kmin >> (top + 1)) << (top + 1);
        for (int bit = top; bit >= 0; --bit) {
            const unsigned p = tau | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < VPT; ++j) cnt += __builtin_popcountll(__ballot(key[j] >= p));
            if (cnt >= k) tau = p;
        }
    }
    int above = 0;
#pragma unroll
    for (int j = 0; j < VPT; ++j) above += __builtin_popcountll(__ballot(key[j] > tau));
    const int ties = k - above;   // how many of the values equal to tau belong to the answer: those with the lowest indices
    // (2) compaction: index order = j-major, lane-minor
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nsel = 0, seen_eq = 0;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const bool gt = key[j] > tau, eq = key[j] == tau;
        const unsigned long long meq = __ballot(eq);
        const bool take = gt || (eq && seen_eq + __builtin_popcountll(meq & lt) < ties);
        const unsigned long long mt = __ballot(take);
        if (take)
            sel[nsel + __builtin_popcountll(mt & lt)] =
                ((unsigned long long)key[j] << 32) | (unsigned)(~(unsigned)(lane + 64 * j));
        nsel += __builtin_popcountll(mt);
        seen_eq += __builtin_popcountll(meq);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (3) bitonic sort of 64 pairs, descending (empty slots: 0, below every pair: an index complement is never 0 here)
    unsigned long long v = lane < k ? sel[lane] : 0ull;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)v, stride, 64), hi = __shfl_xor((unsigned)(v >> 32), stride, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            const bool up = (lane & size) != 0;          // this block ascends
            const bool lower = (lane & stride) == 0;     // this lane keeps the first of the pair
            const bool keep_max = up ? !lower : lower;   // descending blocks keep the larger value in the lower lane
            v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    if (lane < k) out[lane] = (int)(~(unsigned)v);
}

template <int VPT>
__global__ __launch_bounds__(256) void knn_select_kernel(const float *__restrict__ G, const float *__restrict__ xx,
                                                         int N, long long rows, int k, int32_t *__restrict__ idx)
{
    __shared__ unsigned long long s_sel[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const long long b = row / N;
    const float *g = G + row * N;
    const float *xb = xx + b * N;
    const float nxi = -xb[row - b * N];
    float gv[VPT], xv[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        gv[j] = g[c < N ? c : N - 1];
        xv[j] = xb[c < N ? c : N - 1];
    }
    unsigned key[VPT];   // padding columns: key 0 (below every real value, -inf included: its key is 0x007fffff)
#pragma unroll
    for (int j = 0; j < VPT; ++j) key[j] = (lane + 64 * j) < N ? knn_key((nxi - (-2.0f * gv[j])) - xv[j]) : 0u;
    knn_select_keys<VPT>(key, N, k, s_sel[wave], lane, idx + row * k);
}

// The first graph of the network (C = 3: the coordinates themselves) with NO pairwise matrix: the cloud sits in LDS as
// (x, y, z, |p|^2), a wave forms the 2048 values of its row from it -- the k-ordered fma chain of the matrix kernel's K = 3
// product, term by term: x x', then fma(y, y', .), then fma(z, z', .) -- and selects.  The [B, N, N] Gram matrix (403 MB
// written at 2.9 TB/s by a K = 3 "product" at 0.03 of the matrix peak, then read back) never exists.  One workgroup = 64 rows.
constexpr int KNN3_ROWS = 64;
template <int VPT>
__global__ __launch_bounds__(256) void knn3_select_kernel(const float *__restrict__ x, int N, int k, int32_t *__restrict__ idx)
{
    __shared__ float4 s_pts[64 * VPT];
    __shared__ unsigned long long s_sel[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, r0 = blockIdx.x * KNN3_ROWS;
    const float *xb = x + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float px = xb[3 * i], py = xb[3 * i + 1], pz = xb[3 * i + 2];
        s_pts[i] = make_float4(px, py, pz, (px * px + py * py) + pz * pz);      // (the torch expression of _knn_cl, op by op)
    }
    __syncthreads();
    const int r1 = min(N, r0 + KNN3_ROWS);
    for (int r = r0 + wave; r < r1; r += 4) {
        const float4 q = s_pts[r];
        const float nxi = -q.w;
        unsigned key[VPT];
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            const float4 p = s_pts[c < N ? c : N - 1];
            float g = q.x * p.x;
            g = fmaf(q.y, p.y, g);
            g = fmaf(q.z, p.z, g);
            key[j] = c < N ? knn_key((nxi - (-2.0f * g)) - p.w) : 0u;
        }
        knn_select_keys<VPT>(key, N, k, s_sel[wave], lane, idx + ((size_t)b * N + r) * k);
        __builtin_amdgcn_wave_barrier();   // (the next row rewrites s_sel)
    }
}

// rows (b, n, j): [x[b, idx[b,n,j]] - x[b,n] (C), x[b,n] (C), 0-pad]   (src/dgcnn.py:98-105)
__global__ __launch_bounds__(256) void edge_gather_kernel(const float *__restrict__ x,
                                                          const int32_t *__restrict__ idx, int N, int C, int k,
                                                          int ld, long long total, float *__restrict__ out)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long row = id / ld;
        const int c = (int)(id - row * ld);
        const long long bn = row / k;  // b*N + n
        const long long b = bn / N;
        float val = 0.f;
        if (c < C) val = x[(b * N + idx[row]) * C + c] - x[bn * C + c];
        else if (c < 2 * C) val = x[bn * C + (c - C)];
        out[id] = val;
    }
}

// The same for C % 4 == 0 (the 64-wide layers): 16 bytes per thread, ld / 4 threads per row, rows walked with 32-bit
// arithmetic (the scalar form pays three 64-bit divisions per float: 2.5 TB/s of output)
__global__ __launch_bounds__(256) void edge_gather_rows_kernel(const float *__restrict__ x, const int32_t *__restrict__ idx,
                                                               int N, int C, int k, int ld, int rows,
                                                               float *__restrict__ out)
{
    const int tpr = ld >> 2, rpb = 256 / tpr;
    const int rl = threadIdx.x / tpr;
    if (rl >= rpb) return;
    const int q = (threadIdx.x - rl * tpr) * 4;
    for (int row = blockIdx.x * rpb + rl; row < rows; row += gridDim.x * rpb) {
        const int bn = row / k, b = bn / N;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < C) {
            const float4 a = *reinterpret_cast<const float4 *>(x + ((long long)b * N + idx[row]) * C + q);
            const float4 c = *reinterpret_cast<const float4 *>(x + (long long)bn * C + q);
            v = make_float4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w);
        } else if (q < 2 * C) {
            v = *reinterpret_cast<const float4 *>(x + (long long)bn * C + (q - C));
        }
        *reinterpret_cast<float4 *>(out + (long long)row * ld + q) = v;
    }
}

// autograd of the gather: dx[b, idx] += g[:C];  dx[b, n] += g[C:2C] - g[:C].  One lane per float on purpose: a wave's atomic
// instruction then covers 256 contiguous bytes (full rate).  Measured and dropped: four channels per thread (each instruction
// 64 x 4 bytes at a 16-byte stride: 2.0 ms instead of 0.41), and one thread per point walking its k edges with the centre
// term summed in registers (k + 1 atomics per element instead of 2 k, but k dependent load -> atomic steps: 0.9 ms).
__global__ __launch_bounds__(256) void edge_scatter_kernel(const float *__restrict__ g, int ld,
                                                           const int32_t *__restrict__ idx, int N, int C, int k,
                                                           long long total, float *__restrict__ dx)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long row = id / C;
        const int c = (int)(id - row * C);
        const long long bn = row / k;
        const long long b = bn / N;
        const float gd = g[row * ld + c], gc = g[row * ld + C + c];
        unsafeAtomicAdd(dx + (b * N + idx[row]) * C + c, gd);
        unsafeAtomicAdd(dx + bn * C + c, gc - gd);
    }
}

// The two halves of that autograd as two launches: the neighbour term with one atomic per float (the only part that needs
// them), then the centre term -- k edges of a point summed in registers, plain read-modify-write of dx: half the atomics.
__global__ __launch_bounds__(256) void edge_scatter_neigh_kernel(const float *__restrict__ g, int ld,
                                                                 const int32_t *__restrict__ idx, int N, int C, int k,
                                                                 long long total, float *__restrict__ dx)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long row = id / C;
        const int c = (int)(id - row * C);
        const long long b = row / k / N;
        unsafeAtomicAdd(dx + (b * N + idx[row]) * C + c, g[row * ld + c]);
    }
}

__global__ __launch_bounds__(256) void edge_scatter_centre_kernel(const float *__restrict__ g, int ld, int C, int k,
                                                                  long long total, float *__restrict__ dx)
{
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long bn = id / C;
        const int c = (int)(id - bn * C);
        const float *p = g + bn * k * ld + c;
        float acc = 0.f;
        for (int j = 0; j < k; ++j, p += ld) acc += p[C] - p[0];
        dx[id] += acc;
    }
}

extern "C" {

int prifit_knn_topk(const float *G, const float *xx, int B, int N, int k, int32_t *idx, void *stream)
{
    if (!G || !xx || !idx || B <= 0 || N <= 0 || k <= 0 || k > N || N > 4096) return PRIFIT_EINVAL;
    const long long rows = (long long)B * N;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = as_stream(stream);
    if (k <= 64 && N >= 64) {   // (selection + one sort; the round-per-neighbour form below for the rest)
        if (N <= 1024) hipLaunchKernelGGL((knn_select_kernel<16>), grid, block, 0, st, G, xx, N, rows, k, idx);
        else if (N <= 2048) hipLaunchKernelGGL((knn_select_kernel<32>), grid, block, 0, st, G, xx, N, rows, k, idx);
        else hipLaunchKernelGGL((knn_select_kernel<64>), grid, block, 0, st, G, xx, N, rows, k, idx);
        return prifit_check_launch();
    }
    if (N <= 1024) hipLaunchKernelGGL((knn_topk_kernel<16>), grid, block, 0, st, G, xx, N, rows, k, idx);
    else if (N <= 2048) hipLaunchKernelGGL((knn_topk_kernel<32>), grid, block, 0, st, G, xx, N, rows, k, idx);
    else hipLaunchKernelGGL((knn_topk_kernel<64>), grid, block, 0, st, G, xx, N, rows, k, idx);
    return prifit_check_launch();
}

int prifit_knn3_supported(int N, int k) { return (N >= 64 && N <= 2048 && k >= 1 && k <= 64) ? 1 : 0; }

int prifit_knn3_topk(const float *x, int B, int N, int k, int32_t *idx, void *stream)
{
    if (!x || !idx || B <= 0 || B > 65535 || !prifit_knn3_supported(N, k)) return PRIFIT_EINVAL;
    dim3 grid((unsigned)((N + KNN3_ROWS - 1) / KNN3_ROWS), (unsigned)B), block(256);
    hipStream_t st = as_stream(stream);
    if (N <= 1024) hipLaunchKernelGGL((knn3_select_kernel<16>), grid, block, 0, st, x, N, k, idx);
    else hipLaunchKernelGGL((knn3_select_kernel<32>), grid, block, 0, st, x, N, k, idx);
    return prifit_check_launch();
}

int prifit_edge_gather(const float *x, const int32_t *idx, int B, int N, int C, int k, int ld_out, float *out,
                       void *stream)
{
    if (!x || !idx || !out || B <= 0 || N <= 0 || C <= 0 || k <= 0 || ld_out < 2 * C) return PRIFIT_EINVAL;
    const long long total = (long long)B * N * k * ld_out;
    long long gsz = (total + 255) / 256;
    if (gsz > 256 * 32) gsz = 256 * 32;
    const long long rows = (long long)B * N * k;
    if ((C & 3) == 0 && (ld_out & 3) == 0 && ld_out <= 1024 && rows < 2147483647LL && !((uintptr_t)x & 15) && !((uintptr_t)out & 15))
        hipLaunchKernelGGL(edge_gather_rows_kernel, dim3((unsigned)gsz), dim3(256), 0, as_stream(stream), x, idx, N, C, k,
                           ld_out, (int)rows, out);
    else
        hipLaunchKernelGGL(edge_gather_kernel, dim3((unsigned)gsz), dim3(256), 0, as_stream(stream), x, idx, N, C, k,
                           ld_out, total, out);
    return prifit_check_launch();
}

int prifit_edge_scatter(const float *gout, int ld_gout, const int32_t *idx, int B, int N, int C, int k, float *dx,
                        void *stream)
{
    if (!gout || !idx || !dx || B <= 0 || N <= 0 || C <= 0 || k <= 0 || ld_gout < 2 * C) return PRIFIT_EINVAL;
    const long long total = (long long)B * N * k * C;
    long long gsz = (total + 255) / 256;
    if (gsz > 256 * 32) gsz = 256 * 32;
    if (k >= 4) {   // (two passes over g pay once the atomics they save outweigh the second read)
        const long long pts = (long long)B * N * C;
        long long g2 = (pts + 255) / 256;
        if (g2 > 256 * 32) g2 = 256 * 32;
        hipLaunchKernelGGL(edge_scatter_neigh_kernel, dim3((unsigned)gsz), dim3(256), 0, as_stream(stream), gout, ld_gout, idx, N,
                           C, k, total, dx);
        hipLaunchKernelGGL(edge_scatter_centre_kernel, dim3((unsigned)g2), dim3(256), 0, as_stream(stream), gout, ld_gout, C, k,
                           pts, dx);
    } else
        hipLaunchKernelGGL(edge_scatter_kernel, dim3((unsigned)gsz), dim3(256), 0, as_stream(stream), gout, ld_gout,
                           idx, N, C, k, total, dx);
    return prifit_check_launch();
}

}  // extern "C"
