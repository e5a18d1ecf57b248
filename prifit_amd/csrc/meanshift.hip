// Mean-shift clustering on the unit hypersphere (src/mean_shift.py): the bandwidth / assignment /
// update kernels around the N x N x D GEMMs (which run on the MFMA GEMM of gemm.hip).
// All kernels are bandwidth- or latency-class: one wave per matrix row, coalesced row reads,
// wave-shuffle reductions.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// k-th smallest entry of every row (torch.topk(dist, k, largest=False)[0][:, -1], mean_shift.py:156-158)
// One wave per row; the row lives in registers (C <= 64*VPT) and a 32-step bisection over the
// order-preserving integer image of the floats finds the exact k-th value.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

template <int VPT, bool VEC = false>
__global__ __launch_bounds__(256) void kth_smallest_kernel(const float *__restrict__ M, long long rows, int C,
                                                           int k, float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *r = M + row * C;
    unsigned key[VPT];
    // which column slot j of a lane holds: the selection does not care, so 16-byte rows are read as float4 (a quarter of the
    // load instructions: columns 4 lane + 256 (j / 4) + (j % 4)); otherwise lane + 64 j
    auto colof = [&](int j) { return VEC ? 4 * lane + 256 * (j >> 2) + (j & 3) : lane + 64 * j; };
    // branch-free loads (clamped column), padding applied afterwards: a predicated load costs the compiler a branch and
    // a full s_waitcnt vmcnt(0) each, which serialised the VPT loads of a row
    float raw[VPT];
    if (VEC) {
#pragma unroll
        for (int q = 0; q < VPT / 4; ++q) {
            const int c = 4 * lane + 256 * q;
            const float4 v = *reinterpret_cast<const float4 *>(r + (c < C ? c : C - 4));   // C % 4 == 0
            raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            raw[j] = r[c < C ? c : C - 1];
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) key[j] = colof(j) < C ? f2key(raw[j]) : 0xffffffffu;  // padding sorts last
    // Bisection on the key value instead of a masked radix step: the k-th smallest key is the largest p with
    // count(key < p) < k, found bit by bit from the top -- ONE compare-and-count per key and round (2 VALU operations)
    // where the masked digit test of the radix select took five, and no candidate bookkeeping (the padding keys
    // 0xffffffff are never below a pivot, they simply never count).  32 rounds of (VPT counts + one wave sum).
    // Early exit: c_lo / c_hi = number of keys below the current lower / upper bound of the answer; once exactly one key
    // lies between them it IS the answer (a row of distinct distances gets there after ~12 of the 32 rounds).
    // The search starts below the highest bit in which the row's smallest and largest key differ: the bits above are the
    // answer's (a row of nearly equal distances -- an untrained embedding -- shares its top ~20 bits, 20 rounds of nothing).
    unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        kmin = min(kmin, key[j]);
        kmax = max(kmax, colof(j) < C ? key[j] : 0u);
    }
    kmin = ~wave_max_u32_dpp(~kmin);
    kmax = wave_max_u32_dpp(kmax);
    if (kmin == kmax) {   // (wave-uniform) one value in the whole row
        if (lane == 0) out[row] = key2f(kmin);
        return;
    }
    // Fast path (round 4) for k <= 256.  The k-th smallest of the m = ceil(k / 64) smallest keys OF EVERY LANE (64 m >= k
    // candidates, a subset of the row) is an upper bound t0 of the answer, so the answer is the k-th smallest among the keys
    // <= t0 -- usually little more than k of the C keys.  Those are compacted into LDS (at most 256) and the bisection runs on
    // four keys per lane instead of VPT: one compare + ballot per key of the row instead of one per key and ROUND.
    // The value is the same (a selection by value: ties do not matter).  Rows with more than 256 keys <= t0 (many exact
    // ties) take the general path below.
    if (k <= 256) {
        __shared__ unsigned s_c[4][256];
        const int wv = threadIdx.x >> 6;
        const int m = (k + 63) >> 6;                       // wave-uniform, 1..4
        unsigned sm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // the lane's m smallest keys, ascending
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            unsigned x = key[j];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < m) { const unsigned lo = min(x, sm[t]); x = max(x, sm[t]); sm[t] = lo; }
            }
        }
        // t0 = k-th smallest of the 64 m candidates: the largest p with count(candidates < p) < k (>= kmin; bits above `top`
        // are shared by every key of the row)
        auto kth_of = [&](const unsigned (&v)[4], int nv, int kk) {
            const int tp = 31 - __builtin_clz(kmin ^ kmax);
            unsigned pre = tp == 31 ? 0u : (kmin >> (tp + 1)) << (tp + 1);
            for (int bit = tp; bit >= 0; --bit) {
                const unsigned p = pre | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < nv) cnt += __builtin_popcountll(__ballot(v[t] < p));
                if (cnt < kk) pre = p;
            }
            return pre;   // = the kk-th smallest value itself (the largest p with fewer than kk values below it)
        };
        const unsigned t0 = kth_of(sm, m, k);
        const unsigned long long ltm = (1ull << lane) - 1ull;
        int n0 = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const bool take = key[j] <= t0;                // (padding keys 0xffffffff: only if t0 is, i.e. never for k <= C)
            const unsigned long long mt = __ballot(take);
            const int pos = n0 + __builtin_popcountll(mt & ltm);
            if (take && pos < 256) s_c[wv][pos] = key[j];
            n0 += __builtin_popcountll(mt);
        }
        if (n0 <= 256) {   // wave-uniform (n0 >= k by construction)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned cv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) cv[t] = (lane + 64 * t) < n0 ? s_c[wv][lane + 64 * t] : 0xffffffffu;
            const unsigned ans = kth_of(cv, 4, k);
            if (lane == 0) out[row] = key2f(ans);
            return;
        }
    }
    const int top = 31 - __builtin_clz(kmin ^ kmax);
    unsigned prefix = top == 31 ? 0u : (kmin >> (top + 1)) << (top + 1);
    int c_lo = 0, c_hi = C;   // keys below the lower / the upper end of the bracket [prefix, prefix + 2^(top+1))
    for (int bit = top; bit >= 0; --bit) {
        const unsigned p = prefix | (1u << bit);
        // one vector compare per key; the count of its 64-bit lane mask and the running sum are SCALAR instructions (they
        // issue beside the vector pipe), and the sum is wave-wide as it stands: no cross-lane reduction per round
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) cnt += __builtin_popcountll(__ballot(key[j] < p));
        if (cnt < k) { prefix = p; c_lo = cnt; }   // fewer than k keys below p: the answer is >= p
        else c_hi = cnt;
        if (c_hi - c_lo == 1) {
            // the single key in [prefix, prefix + 2^bit): smallest key >= prefix
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < VPT; ++j) m = min(m, key[j] >= prefix ? key[j] : 0xffffffffu);
            prefix = ~wave_max_u32_dpp(~m);
            break;
        }
    }
    if (lane == 0) out[row] = key2f(prefix);
}

// ---------------------------------------------------------------------------------------------
// mean-shift update (mean_shift.py:70-82): O = K X, rsum = row sums of K  ->  Mv = O / rsum;
// new = Z + (Mv - Z); out = new / ||new||.  One wave per point.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ms_update_fwd_kernel(const float *__restrict__ O,
                                                            const float *__restrict__ rsum,
                                                            const float *__restrict__ Z, int D, long long rows,
                                                            float *__restrict__ out, float *__restrict__ nrm_o)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *o = O + row * D;
    const float *z = Z + row * D;
    const float dinv = 1.0f / rsum[row];  // D = 1 / sum(K, 1)
    float nv[4];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        float v = 0.f;
        if (c < D) {
            const float zz = z[c];
            const float m = o[c] * dinv - zz;
            v = zz + m;
        }
        nv[j] = v;
        ss += v * v;
    }
    ss = wave_sum_f32(ss);
    const float n = sqrtf(ss);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < D) out[row * D + c] = nv[j] / n;
    }
    if (lane == 0) nrm_o[row] = n;
}

// Backward of the update: given g = dL/d(out) produce dL/dO [rows, D] and dL/d(rsum) [rows].
// dL/dZ through "Z + (Mv - Z)" is exactly 0.
__global__ __launch_bounds__(256) void ms_update_bwd_kernel(const float *__restrict__ g,
                                                            const float *__restrict__ out,
                                                            const float *__restrict__ nrm,
                                                            const float *__restrict__ O,
                                                            const float *__restrict__ rsum, int D,
                                                            long long rows, int rows_per_batch,
                                                            long long gO_batch_stride, float *__restrict__ gO,
                                                            float *__restrict__ grs)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float *gOr = gO + (row / rows_per_batch) * gO_batch_stride + (row % rows_per_batch) * D;
    const float *o = O + row * D;
    const float rinv = 1.0f / rsum[row];
    const float ninv = 1.0f / nrm[row];
    float gg[4], oo[4];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        gg[j] = g[row * D + (c < D ? c : 0)];      // branch-free loads, masked below
        oo[j] = out[row * D + (c < D ? c : 0)];
        if (c >= D) { gg[j] = 0.f; oo[j] = 0.f; }
        dot += gg[j] * oo[j];
    }
    dot = wave_sum_f32(dot);
    float gr = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < D) {
            const float gnew = (gg[j] - oo[j] * dot) * ninv;  // through the normalisation
            const float mv = o[c] * rinv;
            gOr[c] = gnew * rinv;                            // d/dO[c] of O[c]/r
            gr -= gnew * mv;
        }
    }
    gr = wave_sum_f32(gr);
    if (lane == 0) grs[row] = gr * rinv;                     // d/dr of O/r
}

// ---------------------------------------------------------------------------------------------
// nms (mean_shift.py:162-202) on the chord-distance matrix dist = 2 - 2 Z Z^T  [B, N, N]
// ---------------------------------------------------------------------------------------------
// owner[j] = argmin_i dist[i][j] (first minimum).  dist is bitwise symmetric (same fma chain), so the
// column argmin is read as a row argmin; counts[owner] += 1.
__global__ __launch_bounds__(256) void nms_owner_kernel(const float *__restrict__ dist, int N, long long rows,
                                                        int32_t *__restrict__ owner, int32_t *__restrict__ counts)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *r = dist + row * N;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int c0 = lane; c0 < N; c0 += 64 * 8) {  // eight independent loads in flight, then the ordered compares
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = c0 + 64 * j; v[j] = r[c < N ? c : N - 1]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + 64 * j;
            if (c < N && v[j] < best) { best = v[j]; bi = c; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        owner[row] = bi;
        atomicAdd(counts + (row / N) * N + bi, 1);
    }
}

// For every centre u that owns >= 1 point: chosen[u] = argmax_j (dist[u][j] < b ? counts[j] : 0), first max.
__global__ __launch_bounds__(256) void nms_pick_kernel(const float *__restrict__ dist,
                                                       const int32_t *__restrict__ counts,
                                                       const float *__restrict__ bw, int N, long long rows,
                                                       int32_t *__restrict__ flags)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long b = row / N;
    if (counts[row] == 0) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const float *r = dist + row * N;
    const int32_t *cn = counts + b * N;
    const float thr = bw[b];
    int best = 0, bi = 0x7fffffff;
    for (int c0 = lane; c0 < N; c0 += 64 * 8) {  // branch-free loads of both rows, eight columns in flight
        float d[8];
        int n[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = c0 + 64 * j; const int cc = c < N ? c : N - 1; d[j] = r[cc]; n[j] = cn[cc]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + 64 * j;
            const int v = d[j] < thr ? n[j] : 0;
            if (c < N && (v > best || (v == best && c < bi))) { best = v; bi = c; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) flags[b * N + bi] = 1;
}

// The same pick on the bit mask prifit_chord_sym_mask wrote (bit j of row u = dist[u][j] < b): one wave per centre, a lane per
// mask word -- the 32 counts under a word are read whole, columns ascend inside a lane and across the lanes' words.
__global__ __launch_bounds__(256) void nms_pick_mask_kernel(const uint32_t *__restrict__ mask, const int32_t *__restrict__ counts,
                                                            int N, long long rows, int32_t *__restrict__ flags)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long b = row / N;
    if (counts[row] == 0) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int MW = N / 32;
    const uint32_t *mr = mask + row * MW;
    const int32_t *cn = counts + b * N;
    int best = 0, bi = 0x7fffffff;
    for (int w = lane; w < MW; w += 64) {
        const uint32_t bits = mr[w];
        if (bits == 0) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int4 n4 = *reinterpret_cast<const int4 *>(cn + 32 * w + 4 * q);
            const int nv[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = (bits >> (4 * q + e)) & 1u ? nv[e] : 0;
                const int c = 32 * w + 4 * q + e;
                if (v > best || (v == best && c < bi)) { best = v; bi = c; }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    // (no neighbour with a count: the float form's `v == best && c < bi` ends on column 0)
    if (lane == 0) flags[b * N + (best == 0 ? 0 : bi)] = 1;
}

// Ordered compaction of the flagged centre ids (torch.unique sorts ascending): ids[b][0..count) and count[b].
__global__ __launch_bounds__(256) void nms_compact_kernel(const int32_t *__restrict__ flags, int N, int cap,
                                                          int32_t *__restrict__ ids, int32_t *__restrict__ count)
{
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < N; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const bool f = c < N && flags[(long long)b * N + c] != 0;
        const unsigned long long m = __ballot(f);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        if (f && pos < cap) ids[(long long)b * cap + pos] = c;
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) count[b] = s_base;
    for (int k = s_base + threadIdx.x; k < cap; k += 256) ids[(long long)b * cap + k] = 0;
}

// labels[j] = argmax_k <centre_k, x_j> (first max) with centre_k = Zc[ids[k]], x_j = Z[j]; used[b][label] = 1.
// One wave per point; D <= 256.  (Zc == Z: nms(Z, Z, b), the way upstream calls it.)
__global__ __launch_bounds__(256) void nms_labels_kernel(const float *__restrict__ Zc, const float *__restrict__ Z, int N, int D,
                                                         const int32_t *__restrict__ ids,
                                                         const int32_t *__restrict__ count, int cap,
                                                         long long rows, int32_t *__restrict__ labels,
                                                         int32_t *__restrict__ used)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long b = row / N;
    const int lane = threadIdx.x & 63;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = (lane + 64 * j) < D ? Z[row * D + lane + 64 * j] : 0.f;
    const int K = min(count[b], cap);
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        const float *c = Zc + (b * N + ids[b * cap + k]) * D;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (lane + 64 * j) < D ? x[j] * c[lane + 64 * j] : 0.f;
        s = wave_sum_f32(s);
        if (s > best) { best = s; bk = k; }
    }
    if (lane == 0) {
        labels[row] = bk;
        used[b * cap + bk] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// membership (mean_shift.py:230-247): sim = <centre_k, x_j> / b^2 - gmax; e = exp(clamp(sim, -13, 75));
// W[j][k] = e_k / sum_k e_k for k < count, 0 otherwise.  dots [B, N, KM] are raw dot products.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void membership_fwd_kernel(const float *__restrict__ dots,
                                                             const float *__restrict__ bw,
                                                             const float *__restrict__ gmax,
                                                             const int32_t *__restrict__ count, int N, int KM,
                                                             long long rows, float *__restrict__ W)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const long long b = row / N;
    const int K = min(count[b], KM);
    const float b2 = bw[b] * bw[b];
    const float gm = gmax[b];
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
        const float s = dots[row * KM + k] / b2 - gm;
        sum += expf(fminf(fmaxf(s, -13.f), 75.f));
    }
    for (int k = 0; k < KM; ++k) {
        float w = 0.f;
        if (k < K) {
            const float s = dots[row * KM + k] / b2 - gm;
            w = expf(fminf(fmaxf(s, -13.f), 75.f)) / sum;
        }
        W[row * KM + k] = w;
    }
}

// g_dots[j][k] = dL/d(raw dot): (gW_k - sum_k' gW_k' W_k') * W_k * [not clamped] / b^2
__global__ __launch_bounds__(256) void membership_bwd_kernel(const float *__restrict__ gW,
                                                             const float *__restrict__ W,
                                                             const float *__restrict__ dots,
                                                             const float *__restrict__ bw,
                                                             const float *__restrict__ gmax,
                                                             const int32_t *__restrict__ count, int N, int KM,
                                                             long long rows, float *__restrict__ gdots)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const long long b = row / N;
    const int K = min(count[b], KM);
    const float b2 = bw[b] * bw[b];
    const float gm = gmax[b];
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += gW[row * KM + k] * W[row * KM + k];
    for (int k = 0; k < KM; ++k) {
        float v = 0.f;
        if (k < K) {
            const float s = dots[row * KM + k] / b2 - gm;
            if (s >= -13.f && s <= 75.f) v = (gW[row * KM + k] - dot) * W[row * KM + k] / b2;
        }
        gdots[row * KM + k] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// y = normalize(normalize(x)) row-wise (convex_loss.py:41,57 normalises the embedding twice with
// F.normalize: x / max(|x|, 1e-12)) and its autograd, one wave per row, D <= 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_normalize2_fwd_kernel(const float *__restrict__ X, int D, long long rows,
                                                                 float eps, float *__restrict__ Y)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float v[4];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < D ? X[row * D + c] : 0.f;
        ss += v[j] * v[j];
    }
    const float n1 = fmaxf(sqrtf(wave_sum_f32(ss)), eps);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = v[j] / n1; s2 += v[j] * v[j]; }
    const float n2 = fmaxf(sqrtf(wave_sum_f32(s2)), eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < D) Y[row * D + c] = v[j] / n2;
    }
}

// gx = J1^T J2^T g with J(y = x/n): g -> (g - y (g.y)) / n (zero Jacobian of the clamp when the norm is below eps)
__global__ __launch_bounds__(256) void row_normalize2_bwd_kernel(const float *__restrict__ X, const float *__restrict__ G,
                                                                 int D, long long rows, float eps, float *__restrict__ GX)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float x[4], y1[4], y2[4], g[4];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        x[j] = c < D ? X[row * D + c] : 0.f;
        g[j] = c < D ? G[row * D + c] : 0.f;
        ss += x[j] * x[j];
    }
    const float r1 = sqrtf(wave_sum_f32(ss)), n1 = fmaxf(r1, eps);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { y1[j] = x[j] / n1; s2 += y1[j] * y1[j]; }
    const float r2 = sqrtf(wave_sum_f32(s2)), n2 = fmaxf(r2, eps);
    float d2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { y2[j] = y1[j] / n2; d2 += g[j] * y2[j]; }
    d2 = r2 > eps ? wave_sum_f32(d2) : 0.f;
    float d1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { g[j] = (g[j] - y2[j] * d2) / n2; d1 += g[j] * y1[j]; }
    d1 = r1 > eps ? wave_sum_f32(d1) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < D) GX[row * D + c] = (g[j] - y1[j] * d1) / n1;
    }
}

extern "C" {

int prifit_row_normalize2_fwd(const float *X, int D, long long rows, float eps, float *Y, void *stream)
{
    if (!X || !Y || D <= 0 || D > 256 || rows <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(row_normalize2_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), X, D,
                       rows, eps, Y);
    return prifit_check_launch();
}

int prifit_row_normalize2_bwd(const float *X, const float *G, int D, long long rows, float eps, float *GX, void *stream)
{
    if (!X || !G || !GX || D <= 0 || D > 256 || rows <= 0) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(row_normalize2_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), X, G,
                       D, rows, eps, GX);
    return prifit_check_launch();
}

int prifit_kth_smallest_rows(const float *M, long long rows, int C, int k, float *out, void *stream)
{
    if (!M || !out || rows <= 0 || C <= 0 || k < 1 || k > C || C > 4096) return PRIFIT_EINVAL;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = as_stream(stream);
    const bool vec = (C % 4) == 0 && ((uintptr_t)M & 15) == 0;     // 16-byte rows: float4 loads
#define KTH(V)                                                                                              \
    do {                                                                                                    \
        if (vec) hipLaunchKernelGGL((kth_smallest_kernel<V, true>), grid, block, 0, st, M, rows, C, k, out); \
        else hipLaunchKernelGGL((kth_smallest_kernel<V, false>), grid, block, 0, st, M, rows, C, k, out);    \
    } while (0)
    if (C <= 512) KTH(8);
    else if (C <= 1024) KTH(16);
    else if (C <= 2048) KTH(32);
    else KTH(64);
#undef KTH
    return prifit_check_launch();
}

int prifit_meanshift_update_fwd(const float *O, const float *rowsum, const float *Z, int D, long long rows,
                                float *out, float *nrm, void *stream)
{
    if (!O || !rowsum || !Z || !out || !nrm || rows <= 0 || D <= 0 || D > 256) return PRIFIT_EINVAL;
    hipLaunchKernelGGL(ms_update_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), O,
                       rowsum, Z, D, rows, out, nrm);
    return prifit_check_launch();
}

int prifit_meanshift_update_bwd(const float *g, const float *out, const float *nrm, const float *O,
                                const float *rowsum, int D, int B, int N, float *gO, long long gO_batch_stride,
                                float *g_rowsum, void *stream)
{
    if (!g || !out || !nrm || !O || !rowsum || !gO || !g_rowsum || B <= 0 || N <= 0 || D <= 0 || D > 256 ||
        gO_batch_stride < (long long)N * D)
        return PRIFIT_EINVAL;
    const long long rows = (long long)B * N;
    hipLaunchKernelGGL(ms_update_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), g,
                       out, nrm, O, rowsum, D, rows, N, gO_batch_stride, gO, g_rowsum);
    return prifit_check_launch();
}

// owner / counts from the keys the chord kernel left (prifit_chord_sym_f32 with owner_key): low word = the owner
__global__ __launch_bounds__(256) void nms_owner_from_keys_kernel(const unsigned long long *__restrict__ okey, int N,
                                                                  long long rows, int32_t *__restrict__ owner,
                                                                  int32_t *__restrict__ counts)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    int bi = (int)(unsigned)(okey[row] & 0xffffffffull);
    bi = bi < 0 ? 0 : (bi >= N ? N - 1 : bi);
    owner[row] = bi;
    atomicAdd(counts + (row / N) * N + bi, 1);
}

// counts, flags [rows] and used [B cap] start at zero: one memset when the caller allocated them back to back
static int nms_zero(int32_t *counts, int32_t *flags, int32_t *used, long long rows, size_t nused, hipStream_t st)
{
    if (flags == counts + rows && used == flags + rows)
        return hipMemsetAsync(counts, 0, sizeof(int32_t) * (2 * (size_t)rows + nused), st) == hipSuccess ? PRIFIT_OK : PRIFIT_ELAUNCH;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * rows, st) != hipSuccess) return PRIFIT_ELAUNCH;
    if (hipMemsetAsync(flags, 0, sizeof(int32_t) * rows, st) != hipSuccess) return PRIFIT_ELAUNCH;
    if (hipMemsetAsync(used, 0, sizeof(int32_t) * nused, st) != hipSuccess) return PRIFIT_ELAUNCH;
    return PRIFIT_OK;
}

int prifit_nms(const float *dist, const float *Z, const float *bw, int B, int N, int D, int cap,
               const unsigned long long *owner_key, int32_t *owner, int32_t *counts, int32_t *flags, int32_t *ids,
               int32_t *count, int32_t *labels, int32_t *used, void *stream)
{
    if (!dist || !Z || !bw || !owner || !counts || !flags || !ids || !count || !labels || !used || B <= 0 ||
        N <= 0 || D <= 0 || D > 256 || cap <= 0)
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const long long rows = (long long)B * N;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (nms_zero(counts, flags, used, rows, (size_t)B * cap, st) != PRIFIT_OK) return PRIFIT_ELAUNCH;
    if (owner_key)
        hipLaunchKernelGGL(nms_owner_from_keys_kernel, dim3((unsigned)((rows + 255) / 256)), block, 0, st, owner_key, N, rows,
                           owner, counts);
    else
        hipLaunchKernelGGL(nms_owner_kernel, grid, block, 0, st, dist, N, rows, owner, counts);
    hipLaunchKernelGGL(nms_pick_kernel, grid, block, 0, st, dist, counts, bw, N, rows, flags);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(B), block, 0, st, flags, N, cap, ids, count);
    hipLaunchKernelGGL(nms_labels_kernel, grid, block, 0, st, Z, Z, N, D, ids, count, cap, rows, labels, used);
    return prifit_check_launch();
}

int prifit_nms_mask(const uint32_t *mask, const float *Z, int B, int N, int D, int cap, const unsigned long long *owner_key,
                    int32_t *owner, int32_t *counts, int32_t *flags, int32_t *ids, int32_t *count, int32_t *labels, int32_t *used,
                    void *stream)
{
    if (!mask || !Z || !owner_key || !owner || !counts || !flags || !ids || !count || !labels || !used || B <= 0 || N <= 0 || (N % 128) ||
        D <= 0 || D > 256 || cap <= 0 || ((uintptr_t)counts & 15))
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const long long rows = (long long)B * N;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (nms_zero(counts, flags, used, rows, (size_t)B * cap, st) != PRIFIT_OK) return PRIFIT_ELAUNCH;
    hipLaunchKernelGGL(nms_owner_from_keys_kernel, dim3((unsigned)((rows + 255) / 256)), block, 0, st, owner_key, N, rows, owner, counts);
    hipLaunchKernelGGL(nms_pick_mask_kernel, grid, block, 0, st, mask, counts, N, rows, flags);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(B), block, 0, st, flags, N, cap, ids, count);
    hipLaunchKernelGGL(nms_labels_kernel, grid, block, 0, st, Z, Z, N, D, ids, count, cap, rows, labels, used);
    return prifit_check_launch();
}

int prifit_nms_pair(const float *dist_xc, const float *dist_cc, const float *C, const float *X, const float *bw, int B, int N,
                    int D, int cap, int32_t *owner, int32_t *counts, int32_t *flags, int32_t *ids, int32_t *count,
                    int32_t *labels, int32_t *used, void *stream)
{
    if (!dist_xc || !dist_cc || !C || !X || !bw || !owner || !counts || !flags || !ids || !count || !labels || !used ||
        B <= 0 || N <= 0 || D <= 0 || D > 256 || cap <= 0)
        return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const long long rows = (long long)B * N;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (nms_zero(counts, flags, used, rows, (size_t)B * cap, st) != PRIFIT_OK) return PRIFIT_ELAUNCH;
    // owner[j] = argmin_i (2 - 2 <c_i, x_j>): row j of the points x centres matrix (the column of upstream's centres x
    // points matrix -- the same products in the same k order, so the same bits)
    hipLaunchKernelGGL(nms_owner_kernel, grid, block, 0, st, dist_xc, N, rows, owner, counts);
    hipLaunchKernelGGL(nms_pick_kernel, grid, block, 0, st, dist_cc, counts, bw, N, rows, flags);
    hipLaunchKernelGGL(nms_compact_kernel, dim3(B), block, 0, st, flags, N, cap, ids, count);
    hipLaunchKernelGGL(nms_labels_kernel, grid, block, 0, st, C, X, N, D, ids, count, cap, rows, labels, used);
    return prifit_check_launch();
}

int prifit_membership_fwd(const float *dots, const float *bw, const float *gmax, const int32_t *count, int B,
                          int N, int KM, float *W, void *stream)
{
    if (!dots || !bw || !gmax || !count || !W || B <= 0 || N <= 0 || KM <= 0) return PRIFIT_EINVAL;
    const long long rows = (long long)B * N;
    hipLaunchKernelGGL(membership_fwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       as_stream(stream), dots, bw, gmax, count, N, KM, rows, W);
    return prifit_check_launch();
}

int prifit_membership_bwd(const float *gW, const float *W, const float *dots, const float *bw, const float *gmax,
                          const int32_t *count, int B, int N, int KM, float *gdots, void *stream)
{
    if (!gW || !W || !dots || !bw || !gmax || !count || !gdots || B <= 0 || N <= 0 || KM <= 0)
        return PRIFIT_EINVAL;
    const long long rows = (long long)B * N;
    hipLaunchKernelGGL(membership_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       as_stream(stream), gW, W, dots, bw, gmax, count, N, KM, rows, gdots);
    return prifit_check_launch();
}

}  // extern "C"
