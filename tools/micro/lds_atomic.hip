// Micro-benchmark: issue rate of LDS float atomics (ds_add_f32, no return) against plain LDS stores, per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic.hip -o tools/micro/lds_atomic && tools/micro/lds_atomic
// modes: 0 ds_write_b32 distinct addresses, 1 ds_add_f32 distinct banks (lane -> own word), 2 ds_add_f32 stride 65 (distinct banks,
// another row pattern), 3 ds_add_f32 all lanes one address (worst case), 4 ds_add_f32 two-way conflicts (lane / 2),
// 5 ds_add_f32 random-ish rows (lane * 37 % 64) * 65.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int mode, int iters, unsigned long long *out, float *sink)
{
    __shared__ float s[64 * 66];
    for (int i = threadIdx.x; i < 64 * 66; i += 64) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x;
    int a;
    switch (mode) {
    case 0: case 1: a = lane; break;
    case 2: a = lane * 65; break;
    case 3: a = 0; break;
    case 4: a = lane / 2; break;
    default: a = ((lane * 37) % 64) * 65; break;
    }
    const float v = 1.0f + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int off = a + u;          // 16 different words per iteration
            if (mode == 0) s[off] = v;
            else __hip_atomic_fetch_add(&s[off], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = s[lane * 65] + s[lane];
}
int main()
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4 * 64 * 1024);
    const char *names[] = {"ds_write_b32 distinct", "ds_add_f32 lane->own word", "ds_add_f32 stride 65", "ds_add_f32 ONE address", "ds_add_f32 2-way same address", "ds_add_f32 permuted rows"};
    for (int blocks : {1, 1024})
        for (int mode = 0; mode < 6; ++mode) {
            const int iters = 2000;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, mode, iters, d, sink);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, mode, iters, d, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
            printf("blocks %4d  %-32s  %6.1f memtime ticks per wave instruction   (kernel %.3f ms)\n", blocks, names[mode], (double)c / (iters * 16.0), ms);
        }
    return 0;
}
