// Shared helpers for the gfx950 kernels of libprifit_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prifit_hip.h"

#define PRIFIT_WAVE 64

static inline int prifit_check_launch()
{
    return hipGetLastError() == hipSuccess ? PRIFIT_OK : PRIFIT_ELAUNCH;
}

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 64-bit max across the wave (all lanes receive the result).
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), off, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// 32-bit unsigned max across the wave on the DPP network (no LDS crossbar): quad swaps, row rotates, then the two
// row broadcasts of the GFX9 family; the result is read from lane 63 and returned wave-uniform.
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v)
{
#define PRIFIT_DPP_MAX(ctrl, rmask)                                                                          \
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    PRIFIT_DPP_MAX(0xB1, 0xf);   // quad_perm [1,0,3,2]
    PRIFIT_DPP_MAX(0x4E, 0xf);   // quad_perm [2,3,0,1]
    PRIFIT_DPP_MAX(0x124, 0xf);  // row_ror:4
    PRIFIT_DPP_MAX(0x128, 0xf);  // row_ror:8  -> every lane holds the max of its row of 16
    PRIFIT_DPP_MAX(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    PRIFIT_DPP_MAX(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave max
#undef PRIFIT_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// 32-bit integer sum across the wave on the DPP network, wave-uniform result (same sequence as wave_max_u32_dpp)
__device__ __forceinline__ int wave_sum_i32_dpp(int v)
{
#define PRIFIT_DPP_ADD(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)
    PRIFIT_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
    PRIFIT_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
    PRIFIT_DPP_ADD(0x124, 0xf);  // row_ror:4
    PRIFIT_DPP_ADD(0x128, 0xf);  // row_ror:8  -> every lane holds the sum of its row of 16
    PRIFIT_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3 (other rows add the `old` value 0)
    PRIFIT_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef PRIFIT_DPP_ADD
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// fp64 sum across the wave, the total in every lane.  On the DPP network + four row totals read back through SGPRs: the
// `__shfl_xor` butterfly it replaces is 12 ds_bpermute round trips per value one after the other (~1.3 us of the ~6 us a
// bn_finalize launch takes; 53 such launches per training step).  Fixed order: quad, row of 16, (row 0 + row 1) + (row 2 + row 3).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f64<0x124>(v);    // row_ror:4
    v += dpp_f64<0x128>(v);    // row_ror:8  -> every lane holds the sum of its row of 16
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Workgroup -> (shape, block inside the shape) with all blocks of a shape on ONE XCD (workgroups go to the XCDs round-robin):
// the gather passes re-read a shape's per-point tables k times, 0.5 - 1 MB per shape and table against 4 MB of L2 per XCD --
// in launch order every XCD sees every shape and the re-reads miss.  W = blocks per shape; the last B % 8 shapes stay linear.
__device__ __forceinline__ void xcd_shape_block(int wg, int W, int B, int &b, int &within)
{
    const int full = (B >> 3) << 3;
    if (wg < full * W) {
        const int s = wg >> 3;
        b = (wg & 7) + 8 * (s / W);
        within = s % W;
    } else {
        const int r = wg - full * W;
        b = full + r / W;
        within = r % W;
    }
}

// A [P, C] operand that is NOT stored: row r of it is the first-layer pre-activation of a set-abstraction MLP written by
// linearity (models/pointnet_util.py:243-252: conv1([feat_j | xyz_j - c_g]) = U_j - Vc_g with U per point, Vc per centre
// and the bias folded into U), re-formed on load as U[shape(r) * N + idx[r]] - Vc[r / Kg].  U is small (B N C floats, L2 /
// MALL resident) where the rows are hundreds of MB.  Kg (rows per centre) is a multiple of the 64-row tiles: a tile has one
// centre.  NULL idx = no gather.
struct GatherSrc {
    const int32_t *idx;   // [P] point index of every row (first-index padded, as prifit_sa_group_linear_fwd writes them)
    const float *U;       // [B * N, C]
    const float *Vc;      // [B * S, C]
    int N, S, Kg, C;
    unsigned ubytes;      // B * N * C * 4
};

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm "tail": the kernel that PRODUCES a layer's column sums also finalizes them (round 6; before: a [nslab][2][C] slab
// per producer + a bn_finalize / bn_bwd_finalize launch per layer, 50 launches of ~5 us per training step).
//   * every workgroup adds its per-column partial sums to one of BN_TAIL_REPLICAS [2][C] fp64 accumulators (replica = workgroup
//     id mod 32) with agent-scope atomics.  Measured in the step (profiles/r06_bn_tail.txt): atomics on ONE address are
//     performed one after the other at ~50 ns each, so a single accumulator costs (workgroups x 50 ns) -- +36 us for the 768
//     workgroups of pool_bwd_reduce, +45 us for a 3072-tile product -- while 32 replicas cut the chain to workgroups / 32;
//     the alternative -- a slab per workgroup + the last workgroup summing them with agent-scope loads -- adds 60-500 us
//     (tools/micro/bn_tail.hip);
//   * waits until its atomics are acknowledged (s_waitcnt vmcnt(0): on gfx9 no-return atomics count in vmcnt), draws a
//     ticket, and the workgroup that draws the LAST ticket reads the totals (agent-scope loads) and writes the layer's
//     coefficients -- nobody waits for anybody, so nothing can hang.
// The fp64 sum of <= a few thousand fp32 partials is exact unless their exponents spread over > 2^17, so the order in which
// the atomics land does not show in the fp32 coefficients (BatchNorm statistics: never; backward sums: a last-bit event).
// Outside the HIP memory model on purpose (relaxed atomics + s_waitcnt instead of a release / acquire pair, whose L2
// write-back + invalidate costs ~77 us per launch: csrc/meanshift_rows.hip): valid on gfx942 / gfx950, where agent-scope
// (sc1) accesses are performed at the device coherence point -- hence the guard.
// ---------------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "bn_tail / last-ticket hand-over relies on gfx942 / gfx950 agent-scope (sc1) semantics: revisit before building for another target"
#endif

// The coefficient arithmetic of a BatchNorm layer from its column sums, shared by the finalize kernels (csrc/bn.hip) and the
// tails.  Contraction is switched OFF inside these two functions (`#pragma clang fp contract(off)`; HIP's `__dmul_rn` & co. are
// plain operators that the compiler may still fuse): device code is built with -ffp-contract=fast except the index-critical
// translation units, and whether `q / n - mean * mean` became a fused multiply-add depended on the surrounding kernel -- which
// showed as one-ulp differences in the running variance between two arms of the same layer.
// out: [4][C] at stride C = scale, shift, mean, invstd (forward) / [5][C] = dgamma, dbeta, a, b, d (backward).
__device__ __forceinline__ void bn_fwd_coefs(double s, double q, double count, float gamma, float beta, float eps, float momentum,
                                             float *rmean, float *rvar, float *out, int C)
{
#pragma clang fp contract(off)
    const double mean = s / count;
    const double msq = mean * mean;
    double var = q / count - msq;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma * invstd;
    const float msc = (float)mean * sc;
    out[0] = sc;
    out[C] = beta - msc;
    out[2 * C] = (float)mean;
    out[3 * C] = invstd;
    if (rmean) {
        const double vc = var * count;
        const double unbiased = count > 1.0 ? vc / (count - 1.0) : var;
        const float keep = 1.f - momentum;
        const float km = keep * *rmean, mm = momentum * (float)mean;
        const float kv = keep * *rvar, mv = momentum * (float)unbiased;
        *rmean = km + mm;
        *rvar = kv + mv;
    }
}

//   train: dY = gamma*invstd * (Gm - m1/n - yhat * m2/n)        eval: dY = gamma*invstd_running * Gm  (b = d = 0)
__device__ __forceinline__ void bn_bwd_coefs(double m1, double m2, double count, int training, float scale, float mean, float invstd,
                                             float *out, int C)
{
#pragma clang fp contract(off)
    out[0] = (float)m2;           // dgamma
    out[C] = (float)m1;           // dbeta
    const double a = (double)scale;  // gamma * invstd
    out[2 * C] = (float)a;
    if (training) {
        const double is = (double)invstd, mu = (double)mean;
        const double m2n = m2 / count, m1n = m1 / count;
        const double ais = a * is;
        const double b = ais * m2n;
        const double muis = mu * is;
        const double t = muis * m2n;
        const double d = a * (t - m1n);
        out[3 * C] = (float)(-b);
        out[4 * C] = (float)d;
    } else {
        out[3 * C] = 0.f;
        out[4 * C] = 0.f;
    }
}

constexpr int BN_TAIL_REPLICAS = 32;   // power of two; prifit_bn_tail_floats() tells the caller how much zeroed memory that is

struct BnTail {
    double *acc;                 // [BN_TAIL_REPLICAS][2][C] accumulators, zero on entry (left zero again); NULL = no tail
    int *ticket;                 // zero on entry (left zero again)
    int kind;                    // 1: forward statistics (sum y, sum y^2)   2: backward sums (m1 = sum dY-ish, m2 = sum dY * yhat)
    int C;
    double count;                // rows behind the sums
    const float *p0, *p1, *p2;   // kind 1: gamma, beta, -          kind 2: scale (= gamma * invstd), mean, invstd
    float *rmean, *rvar;         // kind 1: running statistics to update, or NULL
    float eps, momentum;         // kind 1
    int training;                // kind 2
    float *out;                  // kind 1: [4][C] = scale, shift, mean, invstd     kind 2: [5][C] = dgamma, dbeta, a, b, d
};

__device__ __forceinline__ void bn_tail_add(const BnTail &t, int which, int c, float v)
{
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    double *a = t.acc + (size_t)(wg & (BN_TAIL_REPLICAS - 1)) * 2 * t.C;
    __hip_atomic_fetch_add(a + which * t.C + c, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The last workgroup's work: the 32 replicas of every (column, statistic) summed in ascending order -> the layer's coefficients.
// An agent-scope load is a ~2 us round trip, so ALL loads of a pass are in flight at once: four adjacent lanes own one column --
// lane & 1 = which half of the replicas (16 loads each), lane & 2 = which statistic -- and combine through two lane swaps
// (low half + high half, the same order in every run).  4 C work items: one pass for C <= 128 at 512 threads, two at 256.
// Inlined (a real call gave every producer a 0.5 KB-per-lane scratch frame); 32 VGPRs of loads fit the registers the producers'
// dead accumulators leave (GEMMs at their 128-VGPR caps).  The thread count is a multiple of 64 in every producer.
__device__ __forceinline__ void bn_tail_coefficients(const BnTail &t, int tid, int nthreads)
{
    static_assert(BN_TAIL_REPLICAS == 32, "two halves of 16 replicas");
    const int C = t.C, items = 4 * C;
    for (int i0 = 0; i0 < items; i0 += nthreads) {
        const int i = i0 + tid;
        const bool ok = i < items;
        const int ii = ok ? i : 0;
        const int col = ii >> 2, stat = (ii >> 1) & 1, half = ii & 1;
        double *a = t.acc + ((size_t)(half * 16) * 2 + stat) * C + col;
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __hip_atomic_load(a + (size_t)j * 2 * C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) part += v[j];
        if (ok) {
#pragma unroll
            for (int j = 0; j < 16; ++j) __hip_atomic_store(a + (size_t)j * 2 * C, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double oth = __shfl_xor(part, 1, 64);
        const double tot = half == 0 ? part + oth : oth + part;      // replicas 0..15, then 16..31
        const double q = __shfl_xor(tot, 2, 64);                     // lanes with stat == 0: the sum of squares / m2
        if (ok && (i & 3) == 0) {
            if (t.kind == 1) bn_fwd_coefs(tot, q, t.count, t.p0[col], t.p1[col], t.eps, t.momentum, t.rmean ? t.rmean + col : nullptr,
                                          t.rvar ? t.rvar + col : nullptr, t.out + col, C);
            else bn_bwd_coefs(tot, q, t.count, t.training, t.p0[col], t.p1[col], t.p2[col], t.out + col, C);
        }
    }
}

// Called by EVERY thread of EVERY workgroup of the launch exactly once, after the workgroup's bn_tail_add calls (workgroup-
// uniform control flow: it contains barriers).  s_flag: one int of LDS owned by the caller.
__device__ __forceinline__ void bn_tail_finish(const BnTail &t, int *s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nthreads = blockDim.x * blockDim.y * blockDim.z;
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    if (tid == 0) {
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int last = __hip_atomic_fetch_add(t.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1;
        if (last) __hip_atomic_store(t.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    if (*s_flag) bn_tail_coefficients(t, tid, nthreads);
}

// host side: the C-ABI descriptors of include/prifit_hip.h -> the device struct (NULL descriptor = no tail)
static inline BnTail bn_tail_fwd(const prifit_bn_fwd *d, int C)
{
    BnTail t{};
    if (!d || !d->acc) return t;
    t.acc = d->acc; t.ticket = d->ticket; t.kind = 1; t.C = C; t.count = d->count; t.p0 = d->gamma; t.p1 = d->beta;
    t.rmean = d->running_mean; t.rvar = d->running_var; t.eps = d->eps; t.momentum = d->momentum; t.out = d->out;
    return t;
}

static inline BnTail bn_tail_bwd(const prifit_bn_bwd *d, int C)
{
    BnTail t{};
    if (!d || !d->acc) return t;
    t.acc = d->acc; t.ticket = d->ticket; t.kind = 2; t.C = C; t.count = d->count; t.p0 = d->scale; t.p1 = d->mean; t.p2 = d->invstd;
    t.training = d->training; t.out = d->out;
    return t;
}

static inline bool bn_fwd_bad(const prifit_bn_fwd *d)
{
    return d && d->acc && (!d->ticket || !d->gamma || !d->beta || !d->out || !(d->count > 0) || ((uintptr_t)d->acc & 7) ||
                           ((d->running_mean == nullptr) != (d->running_var == nullptr)));
}

static inline bool bn_bwd_bad(const prifit_bn_bwd *d)
{
    return d && d->acc && (!d->ticket || !d->scale || !d->mean || !d->invstd || !d->out || !(d->count > 0) || ((uintptr_t)d->acc & 7));
}
