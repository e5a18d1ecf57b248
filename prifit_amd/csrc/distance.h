// Bit-exact distance recipes shared by the index kernels (pointops.hip, sa_group.hip).  Sources that include
// this header are compiled with -ffp-contract=off; every operation is an explicit __f*_rn intrinsic so that the
// rounding sequence equals the reference's PyTorch-CPU arithmetic (oracle/prifit_oracle.c states the same recipes).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float norm2_3(float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// models/pointnet_util.py:37-39: -2 * (K=3 fma-chain dot) + |src|^2 + |dst|^2
__device__ __forceinline__ float sqdist_expanded(float sx, float sy, float sz, float ss, float dx,
                                                 float dy, float dz, float dd)
{
    float t = __fmul_rn(sx, dx);
    t = __fmaf_rn(sy, dy, t);
    t = __fmaf_rn(sz, dz, t);
    return __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, t), ss), dd);
}

