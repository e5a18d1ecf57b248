// Micro-benchmark (round 6): what does it cost a producer kernel to leave its BatchNorm column statistics in a FINAL form?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bn_tail.hip -o tools/micro/bn_tail && tools/micro/bn_tail
// Every workgroup spins for a fixed time (the "producer"), then
//   mode 0: writes its [2][C] slab (today's form; a second launch reduces the nslab slabs)
//   mode 1: adds its 2 C values to ONE [2][C] fp64 accumulator with agent-scope atomics (all workgroups, the same addresses)
//   mode 2: mode 0 + last-ticket: the workgroup that draws the last ticket sums all nslab slabs (agent-scope loads) in slab order
//   mode 3: mode 1 + last-ticket: the last workgroup reads the 2 C totals
//   mode 4: two-level tickets: groups of 32 slabs summed by their last workgroup into level-2 slabs, the last group sums those
// plus the stand-alone reduce launch (one workgroup per channel over the slabs: bn_finalize's shape).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void st_agent(float *p, float x)
{
    __hip_atomic_store(reinterpret_cast<int *>(p), __builtin_bit_cast(int, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float *p)
{
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<int *>(const_cast<float *>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int MODE>
__global__ __launch_bounds__(256) void producer(int C, int spin, float *slab, double *acc, int *ticket, float *slab2, int *ticket2, double *out)
{
    __shared__ int s_last;
    const int tid = threadIdx.x, nslab = gridDim.x, b = blockIdx.x;
    // the "work": ~spin cycles of dependent VALU
    float v = tid * 1e-3f + b;
    for (int i = 0; i < spin; ++i) v = fmaf(v, 1.0000001f, 1e-7f);
    const int n2 = 2 * C;
    if (MODE == 0 || MODE == 2 || MODE == 4) {
        for (int c = tid; c < n2; c += 256) st_agent(slab + (size_t)b * n2 + c, v + c);
    } else {
        for (int c = tid; c < n2; c += 256) __hip_atomic_fetch_add(acc + c, (double)(v + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 0 || MODE == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 4) {
        const int g = b / 32, ng = (nslab + 31) / 32, members = min(32, nslab - g * 32);
        if (tid == 0) {
            const int last = __hip_atomic_fetch_add(ticket2 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
            if (last) __hip_atomic_store(ticket2 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        for (int c = tid; c < n2; c += 256) {
            float w[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) w[i] = i < members ? ld_agent(slab + (size_t)(g * 32 + i) * n2 + c) : 0.f;
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 32; ++i) s += (double)w[i];
            st_agent(slab2 + (size_t)g * n2 + c, (float)s);     // (a real kernel keeps fp64 here: two floats)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
            if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        for (int c = tid; c < n2; c += 256) {
            double s = 0.0;
            for (int i0 = 0; i0 < ng; i0 += 32) {
                float w[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) w[i] = i0 + i < ng ? ld_agent(slab2 + (size_t)(i0 + i) * n2 + c) : 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) s += (double)w[i];
            }
            out[c] = s;
        }
        return;
    }
    if (tid == 0) {
        const int last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nslab - 1;
        if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (MODE == 2) {
        for (int c = tid; c < n2; c += 256) {
            double s = 0.0;
            for (int i0 = 0; i0 < nslab; i0 += 32) {
                float w[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) w[i] = i0 + i < nslab ? ld_agent(slab + (size_t)(i0 + i) * n2 + c) : 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) s += (double)w[i];
            }
            out[c] = s;
        }
    } else {
        for (int c = tid; c < n2; c += 256) {
            out[c] = __hip_atomic_load(acc + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(acc + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(256) void reduce_launch(const float *slab, int nslab, int C, double *out)
{
    __shared__ double s_s[4];
    for (int st = 0; st < 2; ++st) {
        const int c = blockIdx.x;
        double s = 0.0;
        for (int i = threadIdx.x; i < nslab; i += 256) s += (double)slab[((size_t)i * 2 + st) * C + c];
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) s_s[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[st * C + c] = s_s[0] + s_s[1] + s_s[2] + s_s[3];
        __syncthreads();
    }
}

template <int MODE>
float run(int nslab, int C, int spin, float *slab, double *acc, int *ticket, float *slab2, int *ticket2, double *out, bool with_reduce)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemsetAsync(acc, 0, 8 * 2 * C, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) {
            hipLaunchKernelGGL(producer<MODE>, dim3(nslab), dim3(256), 0, 0, C, spin, slab, acc, ticket, slab2, ticket2, out);
            if (with_reduce) hipLaunchKernelGGL(reduce_launch, dim3(C), dim3(256), 0, 0, slab, nslab, C, out);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms / 20 < best) best = ms / 20;
    }
    return best * 1e3f;
}

int main()
{
    float *slab, *slab2; double *acc, *out; int *ticket, *ticket2;
    hipMalloc(&slab, 4 * 2048 * 2 * 256); hipMalloc(&slab2, 4 * 64 * 2 * 256); hipMalloc(&acc, 8 * 512); hipMalloc(&out, 8 * 512);
    hipMalloc(&ticket, 4); hipMalloc(&ticket2, 4 * 64); hipMemset(ticket, 0, 4); hipMemset(ticket2, 0, 4 * 64);
    for (int spin : {2000, 40000})
        for (int nslab : {256, 512, 1024})
            for (int C : {64, 128, 256}) {
                const float base = run<0>(nslab, C, spin, slab, acc, ticket, slab2, ticket2, out, false);
                const float two = run<0>(nslab, C, spin, slab, acc, ticket, slab2, ticket2, out, true);
                const float at = run<1>(nslab, C, spin, slab, acc, ticket, slab2, ticket2, out, false);
                const float lt = run<2>(nslab, C, spin, slab, acc, ticket, slab2, ticket2, out, false);
                const float alt = run<3>(nslab, C, spin, slab, acc, ticket, slab2, ticket2, out, false);
                const float l2 = run<4>(nslab, C, spin, slab, acc, ticket, slab2, ticket2, out, false);
                printf("spin %5d nslab %4d C %3d | slabs only %7.1f us | + reduce launch %7.1f | fp64 atomics %7.1f | slabs + last-ticket sum %7.1f | "
                       "atomics + last-ticket %7.1f | two-level tickets %7.1f\n", spin, nslab, C, base, two, at, lt, alt, l2);
            }
    return 0;
}
