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

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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
