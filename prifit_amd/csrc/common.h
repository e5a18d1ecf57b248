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
