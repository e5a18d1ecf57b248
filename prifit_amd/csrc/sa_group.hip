// Set-abstraction front end in ONE launch per layer (models/pointnet_util.py:87-107 query_ball_point with
// :19-40 square_distance, :43-60 index_points, :127-133 / :243-249 grouping, and the first 1x1 conv of the
// per-group MLP :195-197 / :250-252) for gfx950.
//
// One wave = one query centre.  The workgroup keeps the shape's cloud in LDS (float4: x, y, z, |p|^2).
//   phase 1  ball query for all radii of the layer in one scan (same two-stage ordered compaction and the same
//            bit-exact expanded-form distances as ball_query_kernel in pointops.hip); the index lists stay in
//            LDS and are copied to HBM with coalesced stores (the backward pass needs them);
//   phase 2  emission: for every (radius, sample) the pre-activation row of the FIRST MLP layer is written
//            straight from the wave, together with the column sums / sums of squares the BatchNorm needs:
//              MODE 0 "direct"  y = W [rel_xyz | feat_j] + b computed from the LDS cloud with the weights in
//                               registers (narrow inputs: SA1, 3..9 input channels) -- no grouped tensor, no GEMM;
//              MODE 1 "gather"  y = U_j - Vc_g + b, the layer by linearity (U = [feat | xyz] W^T per point,
//                               Vc = c Wx^T per centre): rows of U are gathered from L2.
// A row of C floats is written by C/4 lanes as float4, 64/(C/4) rows per wave instruction: every store
// instruction covers 1 KiB of contiguous output.  HBM traffic = the compulsory bytes: clouds + index lists +
// the C-wide rows once (+ U/Vc reads in gather mode); this is the bandwidth-bound kernel of the grouping stage.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "distance.h"

namespace {

constexpr int SG_TILE = 2048;   // cloud size limit (LDS copy)
constexpr int SG_LIST = 320;    // sum of nsample over the radii of one launch
constexpr int SG_CMAX = 128;    // widest first layer
constexpr int SG_UNR = 4;       // independent row chains in flight per lane

struct SAGroupArgs {
    const float *xyz, *new_xyz, *feat;
    int B, N, S, feat_first;
    int feat_xyz;          // MODE 0: the first 3 feature channels ARE the coordinates (taken from the LDS cloud)
    float r2[4];
    int K[4], C[4];
    const float *W[4];     // MODE 0: upstream conv weight [C][D+3]
    const float *U[4];     // MODE 1: [B,N,C]
    const float *Vc[4];    // MODE 1: [B,S,C]
    const float *bias[4];  // [C] or NULL
    float *Y[4];           // [B*S*K, C]
    float *slab[4];        // [B*ceil(S/NW)][2][C] or NULL
    BnTail tail[4];        // per radius: the column statistics finalized by this launch instead of written to slab[r] (common.h)
    int32_t *idx[4];       // [B,S,K]
};

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4g(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
// streaming store: the rows are consumed by a later launch, hundreds of MB of traffic away
__device__ __forceinline__ void st4nt(float *p, float4 v)
{
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(p));
}

// R radii, MODE (0 direct / 1 gather), D feature channels of the direct mode (0, 3 or 6), FX: features 0..2 are the
// coordinates (no global loads for them), NW waves per workgroup
template <int R, int MODE, int D, bool FX, int NW>
__global__ __launch_bounds__(NW * 64) void sa_group_kernel(const SAGroupArgs a)
{
    constexpr int KP = D + 3;
    __shared__ float4 s_pts[SG_TILE];
    __shared__ __attribute__((aligned(16))) float2 s_cand[NW][128];
    __shared__ int s_list[NW][SG_LIST];
    // the statistics staging area reuses the candidate rings: a wave's ring is dead once its phase 1 is over
    static_assert(sizeof(float2) * 128 == sizeof(float) * 2 * SG_CMAX, "ring and staging row have the same size");
    float (*s_red)[2][SG_CMAX] = reinterpret_cast<float (*)[2][SG_CMAX]>(&s_cand[0][0]);

    // XCD-aware placement: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so XCD x works on the
    // shapes x, x+8, ...: the gathered U rows of a shape stay in ONE L2.
    const int N = a.N, S = a.S;
    const int nb = (S + NW - 1) / NW;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int b = (jj / nb) * 8 + xcd, tile = jj % nb;
    __shared__ int s_tail;
    if (b >= a.B) {        // whole workgroup: nothing to do but its tickets
        for (int r = 0; r < R; ++r)
            if (a.tail[r].acc) bn_tail_finish(a.tail[r], &s_tail);
        return;
    }
    const int slab_id = b * nb + tile;

    // (readfirstlane: the wave index, hence the query centre and everything derived from it, is wave-uniform, and the
    // compiler has to know it: the row stores use a per-wave buffer resource)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float *P = a.xyz + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N; i += NW * 64) {
        const float x = P[(size_t)i * 3 + 0], y = P[(size_t)i * 3 + 1], z = P[(size_t)i * 3 + 2];
        s_pts[i] = make_float4(x, y, z, norm2_3(x, y, z));
    }
    __syncthreads();

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int qid = tile * NW + wave;
    const bool valid = qid < S;  // wave-uniform
    const float *Q = a.new_xyz + ((size_t)b * S + (valid ? qid : S - 1)) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    const float qq = norm2_3(qx, qy, qz);
    int *lst = s_list[wave];
    int off[R];
    off[0] = 0;
#pragma unroll
    for (int r = 1; r < R; ++r) off[r] = off[r - 1] + a.K[r - 1];

    // ---------------------------------------------------------------- phase 1: ball query
    if (valid) {
        int cnt[R], first[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { cnt[r] = 0; first[r] = N; }
        float r2max = a.r2[0];
#pragma unroll
        for (int r = 1; r < R; ++r) r2max = fmaxf(r2max, a.r2[r]);
        int fill = 0, done = 0;  // candidates written / consumed (wave-uniform)
        bool finished = false;
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
                const int K = a.K[r];
                if (cnt[r] < K) {
                    const bool pred = act && !(d > a.r2[r]);
                    const unsigned long long m = __ballot(pred);
                    if (m != 0ull) {
                        if (cnt[r] == 0) first[r] = __builtin_amdgcn_readlane(gi, __builtin_ctzll(m));
                        const int pos = cnt[r] + __popcll(m & lt_mask);
                        if (pred && pos < K) lst[off[r] + pos] = gi;
                        cnt[r] += __popcll(m);
                    }
                }
                all = all && cnt[r] >= K;
            }
            done += nproc;
            __builtin_amdgcn_wave_barrier();
            return all;
        };

        for (int c = 0; c < N; c += 64) {
            const int i = c + lane;
            const bool inb = i < N;
            const float4 p = s_pts[inb ? i : 0];
            const float d = sqdist_expanded(qx, qy, qz, qq, p.x, p.y, p.z, p.w);
            const bool pred = inb && !(d > r2max);
            const unsigned long long m = __ballot(pred);
            if (m != 0ull) {
                if (pred) ring[(fill + __popcll(m & lt_mask)) & 127] = make_float2(__int_as_float(i), d);
                fill += __popcll(m);
                if (fill - done >= 64 && consume(64)) { finished = true; break; }
            }
        }
        while (!finished && fill > done) finished = consume(min(64, fill - done));
        // pointnet_util.py:104-106: slots beyond the in-ball count repeat the first in-ball index
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int K = a.K[r];
            for (int k = min(cnt[r], K) + lane; k < K; k += 64) lst[off[r] + k] = first[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int K = a.K[r];
            int32_t *o = a.idx[r] + ((size_t)b * S + qid) * K;
            for (int k = lane; k < K; k += 64) o[k] = lst[off[r] + k];
        }
    }

    // ---------------------------------------------------------------- phase 2: first-layer rows + column statistics
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int K = a.K[r], C = a.C[r];
        const int L = C >> 2, sh = __builtin_ctz(L);  // C is a power of two, 16..128
        const int rpi = 64 >> sh;                     // rows per wave instruction
        const int c4 = lane & (L - 1), rsel = lane >> sh;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        if (valid) {
            const float4 bb = a.bias[r] ? ld4g(a.bias[r] + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int *l = lst + off[r];
            float w[4][KP];
            float4 vc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *Ub = nullptr;
            if (MODE == 0) {
                // x = [rel(3), feat(D)]; upstream column of x[k]: MSG order [feat, rel] (:247), SSG [rel, feat] (:131)
                const float *Wr = a.W[r] + (size_t)(4 * c4) * KP;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        const int col = k < 3 ? (a.feat_first ? D + k : k) : (a.feat_first ? k - 3 : k);
                        w[j][k] = Wr[j * KP + col];
                    }
            } else {
                vc = ld4g(a.Vc[r] + ((size_t)b * S + qid) * C + 4 * c4);
                Ub = a.U[r] + (size_t)b * N * C + 4 * c4;
            }
            const float *Fb = (MODE == 0 && D > 0) ? a.feat + (size_t)b * N * D : nullptr;
            // the rows of this (centre, radius): one buffer resource, the lane's part of the address (its row inside a wave
            // instruction, its 4 channels) is a loop-invariant VGPR -- one 32-bit add per store instead of a 64-bit chain
            const bool rows_out = a.Y[r] != nullptr;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
                rows_out ? a.Y[r] + ((size_t)b * S + qid) * K * C : nullptr, 0, rows_out ? K * C * 4 : 0, 0x00020000);
            const int y_voff = (rsel * C + 4 * c4) * 4;
            const int C4 = C * 4;
            // ALL: K is a whole number of row blocks (every configured layer): no per-row predicates, straight-line code
            auto body = [&](int k0, auto all_tag) {
                constexpr bool ALL = decltype(all_tag)::value;
                bool okk[SG_UNR], inb[SG_UNR];
                int n[SG_UNR];
                float4 p[SG_UNR];
                constexpr int F0 = FX ? 3 : 0;  // first feature channel that is read from HBM / L2
                float f[SG_UNR][D > 0 ? D : 1];
#pragma unroll
                for (int u = 0; u < SG_UNR; ++u) {
                    const int k = k0 + u * rpi + rsel;
                    okk[u] = ALL || k < K;
                    const int nn = l[okk[u] ? k : K - 1];
                    inb[u] = nn >= 0 && nn < N;
                    n[u] = inb[u] ? nn : 0;
                    if (MODE == 0) {
                        p[u] = s_pts[n[u]];
#pragma unroll
                        for (int i = F0; i < D; ++i) f[u][i] = Fb[(size_t)n[u] * D + i];
                        if (FX) { f[u][0] = p[u].x; f[u][1] = p[u].y; f[u][2] = p[u].z; }
                    } else {
                        p[u] = ld4g(Ub + (size_t)n[u] * C);
                    }
                }
#pragma unroll
                for (int u = 0; u < SG_UNR; ++u) {
                    float4 y;
                    if (MODE == 0) {
                        float x[KP];
                        x[0] = inb[u] ? p[u].x - qx : 0.f;
                        x[1] = inb[u] ? p[u].y - qy : 0.f;
                        x[2] = inb[u] ? p[u].z - qz : 0.f;
#pragma unroll
                        for (int i = 0; i < D; ++i) x[3 + i] = inb[u] ? f[u][i] : 0.f;
                        float yy[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float acc = w[j][0] * x[0];
#pragma unroll
                            for (int k = 1; k < KP; ++k) acc = fmaf(w[j][k], x[k], acc);
                            yy[j] = acc;
                        }
                        y = make_float4(yy[0] + bb.x, yy[1] + bb.y, yy[2] + bb.z, yy[3] + bb.w);
                    } else {
                        y = inb[u] ? make_float4(p[u].x - vc.x, p[u].y - vc.y, p[u].z - vc.z, p[u].w - vc.w)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                        y.x += bb.x; y.y += bb.y; y.z += bb.z; y.w += bb.w;
                    }
                    // branch-free: rows beyond K (partial last block) fail the bounds check of the store -- their row goes into
                    // the checked lane offset -- and enter the statistics as zeros
                    const f32x4 t = {y.x, y.y, y.z, y.w};
                    const auto tv = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, t);
                    // (the row block goes into the LANE offset, one v_add per store, not into the scalar offset: with an SGPR
                    // offset hipcc 7.2 leaves out the wait state between a 128-bit buffer store and the next VALU write of its
                    // data registers -- lanes 12-15 of every 16 stored the following row's second dword)
                    if (rows_out)   // (wave-uniform: a scale whose rows every consumer re-forms from U / Vc stores none)
                        __builtin_amdgcn_raw_buffer_store_b128(tv, yrs, y_voff + (k0 + u * rpi) * C4, 0, 2 /* nt */);
                    if (!okk[u]) y = make_float4(0.f, 0.f, 0.f, 0.f);
                    s0.x += y.x; s0.y += y.y; s0.z += y.z; s0.w += y.w;
                    s1.x += y.x * y.x; s1.y += y.y * y.y; s1.z += y.z * y.z; s1.w += y.w * y.w;
                }
            };
            if (K % (rpi * SG_UNR) == 0) {
                for (int k0 = 0; k0 < K; k0 += rpi * SG_UNR) body(k0, std::true_type());
            } else {
                for (int k0 = 0; k0 < K; k0 += rpi * SG_UNR) body(k0, std::false_type());
            }
        }
        const bool tail = a.tail[r].acc != nullptr;
        if (a.slab[r] || tail) {  // block-uniform
            for (int o = L; o < 64; o <<= 1) {
                s0.x += __shfl_xor(s0.x, o, 64); s0.y += __shfl_xor(s0.y, o, 64);
                s0.z += __shfl_xor(s0.z, o, 64); s0.w += __shfl_xor(s0.w, o, 64);
                s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64);
                s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
            }
            if (rsel == 0) {
                *reinterpret_cast<float4 *>(&s_red[wave][0][4 * c4]) = s0;
                *reinterpret_cast<float4 *>(&s_red[wave][1][4 * c4]) = s1;
            }
            __syncthreads();
            for (int t = threadIdx.x; t < 2 * C; t += NW * 64) {
                const int which = t >= C ? 1 : 0, c = t - which * C;
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) s += s_red[wv][which][c];
                if (tail) bn_tail_add(a.tail[r], which, c, s);
                else a.slab[r][((size_t)slab_id * 2 + which) * C + c] = s;
            }
            __syncthreads();
            if (tail) bn_tail_finish(a.tail[r], &s_tail);
        }
    }
}

// dW of the direct-mode first layer: dW[c][col] = sum over grouped samples of dY[row][c] * x_row[col] with
// x_row = [rel_xyz | feat_j] re-formed from the index lists (the grouped tensor is never materialised).
// Every workgroup reduces a contiguous range of rows and writes one [C][D+3] partial (upstream column order).
// BNB (fused BatchNorm backward): dY is not materialised; `dY` then holds G = the gradient w.r.t. relu(bn(Y1)) and
// dy = ca * (Y1*scale+shift > 0 ? g : 0) + cb * y + cd is formed on load from (G, Y1) -- the bn_relu_bwd_apply pass
// that would write dY and the read of it here are gone.
struct DwBn {
    const float *Y1, *scale, *shift, *ca, *cb, *cd;
    const float *U, *Vc;   // Y1 == NULL: the rows are not stored, y1[row] = U[b * N + idx[row]] - Vc[row / K]  (first layer by linearity)
};

template <int D, bool BNB>
__global__ __launch_bounds__(256) void sa_first_layer_dw_kernel(
    const float *__restrict__ dY, const int32_t *__restrict__ idx, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ feat, int N, int S, int K, int C, int feat_first,
    long long P, long long rows_per_block, float *__restrict__ partial, const DwBn bn)
{
    constexpr int KP = D + 3;
    __shared__ float s_x[256][KP];
    __shared__ float s_acc[4][SG_CMAX * KP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = C >> 2, sh = __builtin_ctz(L), rpi = 64 >> sh;
    const int c4 = lane & (L - 1), rsel = lane >> sh;
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    const long long r_end = r_begin + rows_per_block < P ? r_begin + rows_per_block : P;
    float acc[4][KP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[j][k] = 0.f;
    float4 b_s, b_t, b_a, b_b, b_d;
    b_s = b_t = b_a = b_b = b_d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BNB) {
        b_s = ld4g(bn.scale + 4 * c4); b_t = ld4g(bn.shift + 4 * c4);
        b_a = ld4g(bn.ca + 4 * c4); b_b = ld4g(bn.cb + 4 * c4); b_d = ld4g(bn.cd + 4 * c4);
    }

    for (long long base = r_begin; base < r_end; base += 256) {
        {   // x rows of this chunk: one thread per row
            const long long row = base + threadIdx.x;
            float x[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) x[k] = 0.f;
            if (row < r_end) {
                const long long g = row / K;  // b*S + s
                const int bb = (int)(g / S);
                const int n = idx[row];
                if (n >= 0 && n < N) {
                    const float *p = xyz + ((size_t)bb * N + n) * 3;
                    const float *c = new_xyz + (size_t)g * 3;
                    x[0] = p[0] - c[0]; x[1] = p[1] - c[1]; x[2] = p[2] - c[2];
#pragma unroll
                    for (int i = 0; i < D; ++i) x[3 + i] = feat[((size_t)bb * N + n) * D + i];
                }
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) s_x[threadIdx.x][k] = x[k];
        }
        __syncthreads();
        for (int i0 = 0; i0 < 64; i0 += rpi * SG_UNR) {
            float4 gy[SG_UNR];
            int lr[SG_UNR];
#pragma unroll
            for (int u = 0; u < SG_UNR; ++u) {
                lr[u] = wave * 64 + i0 + u * rpi + rsel;
                const long long row = base + lr[u];
                const bool ok = (i0 + u * rpi + rsel) < 64 && row < r_end;
                gy[u] = ld4g(dY + (size_t)(ok ? row : r_begin) * C + 4 * c4);
                if (BNB) {
                    float4 y;
                    if (bn.Y1) {
                        y = ld4g(bn.Y1 + (size_t)(ok ? row : r_begin) * C + 4 * c4);
                    } else {   // (block-uniform branch)
                        const long long rr = ok ? row : r_begin;
                        const long long gq = rr / K;
                        const int nn = idx[rr];
                        const float4 u = ld4g(bn.U + ((size_t)(gq / S) * N + ((nn >= 0 && nn < N) ? nn : 0)) * C + 4 * c4);
                        const float4 v = ld4g(bn.Vc + (size_t)gq * C + 4 * c4);
                        y = make_float4(u.x - v.x, u.y - v.y, u.z - v.z, u.w - v.w);
                    }
                    gy[u].x = fmaf(b_a.x, fmaf(y.x, b_s.x, b_t.x) > 0.f ? gy[u].x : 0.f, fmaf(b_b.x, y.x, b_d.x));
                    gy[u].y = fmaf(b_a.y, fmaf(y.y, b_s.y, b_t.y) > 0.f ? gy[u].y : 0.f, fmaf(b_b.y, y.y, b_d.y));
                    gy[u].z = fmaf(b_a.z, fmaf(y.z, b_s.z, b_t.z) > 0.f ? gy[u].z : 0.f, fmaf(b_b.z, y.z, b_d.z));
                    gy[u].w = fmaf(b_a.w, fmaf(y.w, b_s.w, b_t.w) > 0.f ? gy[u].w : 0.f, fmaf(b_b.w, y.w, b_d.w));
                }
                if (!ok) gy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                lr[u] = ok ? lr[u] : 0;
            }
#pragma unroll
            for (int u = 0; u < SG_UNR; ++u) {
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const float xv = s_x[lr[u]][k];
                    acc[0][k] = fmaf(gy[u].x, xv, acc[0][k]);
                    acc[1][k] = fmaf(gy[u].y, xv, acc[1][k]);
                    acc[2][k] = fmaf(gy[u].z, xv, acc[2][k]);
                    acc[3][k] = fmaf(gy[u].w, xv, acc[3][k]);
                }
            }
        }
        __syncthreads();
    }
    for (int o = L; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[j][k] += __shfl_xor(acc[j][k], o, 64);
    }
    if (rsel == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KP; ++k) s_acc[wave][(4 * c4 + j) * KP + k] = acc[j][k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < C * KP; t += 256) {
        const int c = t / KP, k = t - c * KP;
        const int col = k < 3 ? (feat_first ? D + k : k) : (feat_first ? k - 3 : k);
        partial[(size_t)blockIdx.x * C * KP + c * KP + col] = s_acc[0][t] + s_acc[1][t] + s_acc[2][t] + s_acc[3][t];
    }
}

// The per-point / per-centre tables of the first layer by linearity for narrow inputs (3..9 data channels), all radii of a
// level in ONE launch: U_r[b,n,:] = W_r [feat_n | xyz_n] + b_r, Vc_r[b,s,:] = W_r,x c_s (W_r [C_r][D+3] in upstream column
// order, see sa_group_kernel), so that conv1([feat_j | xyz_j - c]) + b = U_j - Vc.  One thread per (row, 4 channels).
struct SATablesArgs {
    const float *xyz, *new_xyz, *feat;
    int B, N, S, D, feat_first, R;
    int C[4];
    const float *W[4], *bias[4];
    float *U[4], *Vc[4];
};

__global__ __launch_bounds__(256) void sa_point_tables_kernel(const SATablesArgs a)
{
    const int r = blockIdx.y;
    const int C = a.C[r], L = C >> 2, KP = a.D + 3;
    const long long nu = (long long)a.B * a.N, nv = (long long)a.B * a.S;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = id / L;
    const int c4 = (int)(id - row * L);
    if (row >= nu + nv) return;
    const float *Wr = a.W[r] + (size_t)(4 * c4) * KP;
    const int xc = a.feat_first ? a.D : 0, fc = a.feat_first ? 0 : 3;   // columns of rel_xyz / features in W
    float y[4];
    if (row < nu) {
        const float *p = a.xyz + row * 3;
        const float px = p[0], py = p[1], pz = p[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *w = Wr + j * KP;
            float acc = w[xc] * px;
            acc = fmaf(w[xc + 1], py, acc);
            acc = fmaf(w[xc + 2], pz, acc);
            for (int i = 0; i < a.D; ++i) acc = fmaf(w[fc + i], a.feat[row * a.D + i], acc);
            y[j] = acc + (a.bias[r] ? a.bias[r][4 * c4 + j] : 0.f);
        }
        *reinterpret_cast<float4 *>(a.U[r] + row * C + 4 * c4) = make_float4(y[0], y[1], y[2], y[3]);
    } else {
        const long long g = row - nu;
        const float *c = a.new_xyz + g * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *w = Wr + j * KP;
            y[j] = fmaf(w[xc + 2], cz, fmaf(w[xc + 1], cy, w[xc] * cx));
        }
        *reinterpret_cast<float4 *>(a.Vc[r] + g * C + 4 * c4) = make_float4(y[0], y[1], y[2], y[3]);
    }
}

template <int R, int MODE, int D, bool FX>
int launch_nw(const SAGroupArgs &a, hipStream_t st)
{
    // 8 queries per workgroup share one LDS copy of the cloud; 4 when the layer has few queries (fuller last wave of
    // workgroups on 256 CUs)
    const int B8 = (a.B + 7) / 8 * 8;
    if ((long long)a.B * ((a.S + 7) / 8) >= 1024) {
        hipLaunchKernelGGL((sa_group_kernel<R, MODE, D, FX, 8>), dim3(B8 * ((a.S + 7) / 8)), dim3(512), 0, st, a);
    } else {
        hipLaunchKernelGGL((sa_group_kernel<R, MODE, D, FX, 4>), dim3(B8 * ((a.S + 3) / 4)), dim3(256), 0, st, a);
    }
    return prifit_check_launch();
}

template <int MODE, int D, bool FX>
int launch_r(const SAGroupArgs &a, int R, hipStream_t st)
{
    switch (R) {
        case 1: return launch_nw<1, MODE, D, FX>(a, st);
        case 2: return launch_nw<2, MODE, D, FX>(a, st);
        case 3: return launch_nw<3, MODE, D, FX>(a, st);
        default: return launch_nw<4, MODE, D, FX>(a, st);
    }
}

}  // namespace

extern "C" {

int prifit_sa_group_queries_per_slab(int B, int S)
{
    return ((long long)B * ((S + 7) / 8) >= 1024) ? 8 : 4;
}

int prifit_sa_group_linear_fwd(const float *xyz, const float *new_xyz, int B, int N, int S, int R,
                               const float *radius2, const int *nsample, const int *width, int mode,
                               const float *feat, int D, int feat_first, int feat_xyz, const float *const *W,
                               const float *const *U, const float *const *Vc, const float *const *bias,
                               float *const *Y, float *const *slab, int32_t *const *idx, const prifit_bn_fwd *const *bn,
                               void *stream)
{
    if (!xyz || !new_xyz || !radius2 || !nsample || !width || !bias || !Y || (!slab && !bn) || !idx || B <= 0 || N <= 0 ||
        N > SG_TILE || S <= 0 || R < 1 || R > 4 || (mode != 0 && mode != 1))
        return PRIFIT_EINVAL;
    if (mode == 0 && (!W || (D != 0 && D != 3 && D != 6) || (D > 0 && !feat) || (feat_xyz && D < 3)))
        return PRIFIT_EINVAL;
    if (mode == 1 && (!U || !Vc)) return PRIFIT_EINVAL;
    SAGroupArgs a;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feat = feat; a.B = B; a.N = N; a.S = S; a.feat_first = feat_first;
    a.feat_xyz = feat_xyz;
    int ksum = 0;
    for (int r = 0; r < 4; ++r) {
        const bool in = r < R;
        a.r2[r] = in ? radius2[r] : 0.f;
        a.K[r] = in ? nsample[r] : 0;
        a.C[r] = in ? width[r] : 0;
        a.W[r] = (in && mode == 0) ? W[r] : nullptr;
        a.U[r] = (in && mode == 1) ? U[r] : nullptr;
        a.Vc[r] = (in && mode == 1) ? Vc[r] : nullptr;
        a.bias[r] = in ? bias[r] : nullptr;
        a.Y[r] = in ? Y[r] : nullptr;
        a.slab[r] = (in && slab) ? slab[r] : nullptr;
        a.idx[r] = in ? idx[r] : nullptr;
        a.tail[r] = BnTail{};
        if (!in) continue;
        const int C = width[r];
        if (bn && bn[r]) {
            if (bn_fwd_bad(bn[r])) return PRIFIT_EINVAL;
            a.tail[r] = bn_tail_fwd(bn[r], C);
        }
        // (gather mode: Y[r] may be NULL -- index lists and statistics only, the rows are re-formed by their consumers)
        if (nsample[r] < 1 || C < 16 || C > SG_CMAX || (C & (C - 1)) || (!Y[r] && mode == 0) || !idx[r] ||
            (mode == 0 ? !W[r] : (!U[r] || !Vc[r])) || ((uintptr_t)Y[r] & 15) ||
            (mode == 1 && (((uintptr_t)U[r] | (uintptr_t)Vc[r]) & 15)) || (bias[r] && ((uintptr_t)bias[r] & 15)))
            return PRIFIT_EINVAL;
        ksum += nsample[r];
    }
    if (ksum > SG_LIST) return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if (mode == 1) return launch_r<1, 0, false>(a, R, st);
    if (D == 0) return launch_r<0, 0, false>(a, R, st);
    if (D == 3) return feat_xyz ? launch_r<0, 3, true>(a, R, st) : launch_r<0, 3, false>(a, R, st);
    return feat_xyz ? launch_r<0, 6, true>(a, R, st) : launch_r<0, 6, false>(a, R, st);
}

int prifit_sa_point_tables(const float *xyz, const float *new_xyz, const float *feat, int B, int N, int S, int D, int feat_first,
                           int R, const int *width, const float *const *W, const float *const *bias, float *const *U,
                           float *const *Vc, void *stream)
{
    if (!xyz || !new_xyz || !width || !W || !bias || !U || !Vc || B <= 0 || N <= 0 || S <= 0 || D < 0 || D > 9 || (D > 0 && !feat) ||
        R < 1 || R > 4)
        return PRIFIT_EINVAL;
    SATablesArgs a;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feat = feat; a.B = B; a.N = N; a.S = S; a.D = D; a.feat_first = feat_first; a.R = R;
    int lmax = 0;
    for (int r = 0; r < 4; ++r) {
        const bool in = r < R;
        a.C[r] = in ? width[r] : 0; a.W[r] = in ? W[r] : nullptr; a.bias[r] = in ? bias[r] : nullptr;
        a.U[r] = in ? U[r] : nullptr; a.Vc[r] = in ? Vc[r] : nullptr;
        if (!in) continue;
        if (width[r] < 4 || (width[r] & 3) || !W[r] || !U[r] || !Vc[r] || (((uintptr_t)U[r] | (uintptr_t)Vc[r]) & 15))
            return PRIFIT_EINVAL;
        lmax = width[r] / 4 > lmax ? width[r] / 4 : lmax;
    }
    const long long threads = ((long long)B * N + (long long)B * S) * lmax;
    hipLaunchKernelGGL(sa_point_tables_kernel, dim3((unsigned)((threads + 255) / 256), R), dim3(256), 0, as_stream(stream), a);
    return prifit_check_launch();
}

static int dw_launch(const float *dY, const int32_t *idx, const float *xyz, const float *new_xyz, const float *feat,
                     int B, int N, int S, int K, int C, int D, int feat_first, int nblocks, float *partial,
                     const DwBn *bn, void *stream)
{
    if (!dY || !idx || !xyz || !new_xyz || !partial || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C < 16 ||
        C > SG_CMAX || (C & (C - 1)) || (D != 0 && D != 3 && D != 6) || (D > 0 && !feat) || nblocks < 1 ||
        ((uintptr_t)dY & 15))
        return PRIFIT_EINVAL;
    const long long P = (long long)B * S * K;
    long long per = (P + nblocks - 1) / nblocks;
    per = (per + 255) / 256 * 256;
    if ((long long)nblocks * per < P) return PRIFIT_EINVAL;
    hipStream_t st = as_stream(stream);
    DwBn none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#define DW_LAUNCH(DD)                                                                                                \
    if (bn) hipLaunchKernelGGL((sa_first_layer_dw_kernel<DD, true>), dim3(nblocks), dim3(256), 0, st, dY, idx, xyz,     \
                               new_xyz, feat, N, S, K, C, feat_first, P, per, partial, *bn);                          \
    else hipLaunchKernelGGL((sa_first_layer_dw_kernel<DD, false>), dim3(nblocks), dim3(256), 0, st, dY, idx, xyz,      \
                            new_xyz, feat, N, S, K, C, feat_first, P, per, partial, none)
    if (D == 0) { DW_LAUNCH(0); }
    else if (D == 3) { DW_LAUNCH(3); }
    else { DW_LAUNCH(6); }
#undef DW_LAUNCH
    return prifit_check_launch();
}

int prifit_sa_first_layer_dw(const float *dY, const int32_t *idx, const float *xyz, const float *new_xyz,
                             const float *feat, int B, int N, int S, int K, int C, int D, int feat_first,
                             int nblocks, float *partial, void *stream)
{
    return dw_launch(dY, idx, xyz, new_xyz, feat, B, N, S, K, C, D, feat_first, nblocks, partial, nullptr, stream);
}

int prifit_sa_first_layer_dw_bn(const float *G, const float *Y1, const float *scale, const float *shift,
                                const float *coef_a, const float *coef_b, const float *coef_d, const int32_t *idx,
                                const float *xyz, const float *new_xyz, const float *feat, int B, int N, int S, int K,
                                int C, int D, int feat_first, int nblocks, float *partial, void *stream)
{
    if (!Y1 || !scale || !shift || !coef_a || !coef_b || !coef_d || ((uintptr_t)Y1 & 15)) return PRIFIT_EINVAL;
    const DwBn bn = {Y1, scale, shift, coef_a, coef_b, coef_d, nullptr, nullptr};
    return dw_launch(G, idx, xyz, new_xyz, feat, B, N, S, K, C, D, feat_first, nblocks, partial, &bn, stream);
}

int prifit_sa_first_layer_dw_bn_gather(const float *G, const float *U, const float *Vc, const float *scale, const float *shift,
                                       const float *coef_a, const float *coef_b, const float *coef_d, const int32_t *idx,
                                       const float *xyz, const float *new_xyz, const float *feat, int B, int N, int S, int K,
                                       int C, int D, int feat_first, int nblocks, float *partial, void *stream)
{
    if (!U || !Vc || !scale || !shift || !coef_a || !coef_b || !coef_d || (((uintptr_t)U | (uintptr_t)Vc) & 15))
        return PRIFIT_EINVAL;
    const DwBn bn = {nullptr, scale, shift, coef_a, coef_b, coef_d, U, Vc};
    return dw_launch(G, idx, xyz, new_xyz, feat, B, N, S, K, C, D, feat_first, nblocks, partial, &bn, stream);
}

}  // extern "C"
