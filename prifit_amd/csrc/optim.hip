// Adam over the whole parameter set in ONE launch (train_partseg_shapenet.py:252-259: torch.optim.Adam(lr, betas (0.9, 0.999),
// eps 1e-8, weight_decay) and optimizer.step() at :398 / :451).
//
// Parameters, first and second moments live in three flat fp32 buffers with the same layout: parameter s occupies
// [off[s], off[s] + len[s]), off[s] a multiple of 128 floats (every tensor keeps the 512-byte alignment the allocator gave it:
// the GEMM loaders take 16-byte aligned weight rows).  Gradients stay where autograd left them -- G[s] is a device pointer per
// parameter, NULL for a parameter without a gradient (skipped like torch does: no weight decay, no moment decay, no step).
// torch's fused Adam takes three launches for the 144 tensors of the MSG network (kernel-argument space) plus a
// multi-tensor add for the step counters: 0.1 ms of GPU time and ~0.6 ms of host time per step; here one launch, memory-bound
// over 4 x 1.76 M floats.  The arithmetic is torch's single-tensor Adam (torch/optim/adam.py:_single_tensor_adam), fp32 with
// the bias corrections in double like its host code.
#include "common.h"

namespace {

struct AdamArgs {
    float *P, *M, *V;
    const float *const *G;
    const int *off, *len;
    int nparams;
    long long total;           // floats in the flat buffers
    const int *step_in;
    int *step_out;
    float lr, b1, b2, eps, wd;
    const int *skip;           // optional device flag: non-zero = leave everything untouched (a discarded step)
};

__global__ __launch_bounds__(256) void adam_flat_kernel(AdamArgs a)
{
    extern __shared__ int s_off[];          // [nparams + 1]
    for (int i = threadIdx.x; i <= a.nparams; i += 256) s_off[i] = i < a.nparams ? a.off[i] : 0x7fffffff;
    __syncthreads();
    const bool skip = a.skip && *a.skip != 0;
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (base >= a.total) return;
    // segment of this float4: the last s with off[s] <= base (segments start on multiples of 4, so one float4 = one segment)
    int lo = 0, hi = a.nparams;            // invariant: off[lo] <= base < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((long long)s_off[mid] <= base) lo = mid; else hi = mid;
    }
    const int s = lo, local = (int)(base - s_off[s]), n = a.len[s];
    if (local >= n) return;                 // alignment padding behind the tensor
    const float *g = a.G[s];
    const int t_in = a.step_in[s];
    const bool active = g != nullptr && !skip;
    if (local == 0) a.step_out[s] = t_in + (active ? 1 : 0);
    if (!active) return;
    const double t = (double)(t_in + 1);
    const double bc1 = 1.0 - pow((double)a.b1, t), bc2 = 1.0 - pow((double)a.b2, t);
    const float step_size = (float)((double)a.lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const int cnt = min(4, n - local);
    float gv[4] = {0.f, 0.f, 0.f, 0.f};
    if (cnt == 4 && (((uintptr_t)(g + local)) & 15) == 0) {
        const float4 q = *reinterpret_cast<const float4 *>(g + local);
        gv[0] = q.x; gv[1] = q.y; gv[2] = q.z; gv[3] = q.w;
    } else {
        for (int j = 0; j < cnt; ++j) gv[j] = g[local + j];
    }
    float4 p4 = *reinterpret_cast<float4 *>(a.P + base), m4 = *reinterpret_cast<float4 *>(a.M + base),
           v4 = *reinterpret_cast<float4 *>(a.V + base);
    float *p = &p4.x, *m = &m4.x, *v = &v4.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < cnt) {
            const float gj = gv[j] + a.wd * p[j];                       // grad.add(param, alpha=weight_decay)
            m[j] = m[j] + (1.f - a.b1) * (gj - m[j]);                   // exp_avg.lerp_(grad, 1 - beta1)
            v[j] = a.b2 * v[j] + (1.f - a.b2) * gj * gj;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            const float denom = sqrtf(v[j]) / bc2_sqrt + a.eps;
            p[j] = p[j] - step_size * (m[j] / denom);                   // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
    }
    *reinterpret_cast<float4 *>(a.P + base) = p4;
    *reinterpret_cast<float4 *>(a.M + base) = m4;
    *reinterpret_cast<float4 *>(a.V + base) = v4;
}

}  // namespace

extern "C" int prifit_adam_flat_alignment(void) { return 128; }

extern "C" int prifit_adam_flat(float *params, float *exp_avg, float *exp_avg_sq, const float *const *grads, const int32_t *offsets,
                                const int32_t *lengths, int nparams, long long total, const int32_t *step_in, int32_t *step_out,
                                float lr, float beta1, float beta2, float eps, float weight_decay, const int32_t *skip, void *stream)
{
    if (!params || !exp_avg || !exp_avg_sq || !grads || !offsets || !lengths || !step_in || !step_out || step_in == step_out ||
        nparams <= 0 || nparams > 8191 || total <= 0 || (total & 3) || ((uintptr_t)params & 15) || ((uintptr_t)exp_avg & 15) ||
        ((uintptr_t)exp_avg_sq & 15) || !(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
        return PRIFIT_EINVAL;
    AdamArgs a{params, exp_avg, exp_avg_sq, grads, offsets, lengths, nparams, total, step_in, step_out, lr, beta1, beta2, eps,
               weight_decay, skip};
    const long long nvec = total / 4;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), sizeof(int) * (nparams + 1), as_stream(stream), a);
    return prifit_check_launch();
}
