// Weights-stationary streaming fp32 GEMM for the tall-and-skinny layers of the shared per-position MLPs
// (models/pointnet_util.py:195-199, :252-256 and their autograd): C[M,N] = A[M,K] . B^T (NT, forward) or
// A[M,K] . B[K,N] (NN, dA) with M = grouped samples (10^5..10^6) and N, K <= 128.
//
// These products are HBM-bound (the A rows are read once, the C rows written once, 2MNK/(4M(K+N)) < 32 flop/B), so
// the kernel is built around bytes in flight rather than around the matrix cores:
//   * persistent workgroups (a few per CU) walk the 64-row tiles of A with a grid stride;
//   * the whole B operand lives in REGISTERS for the life of the workgroup: wave w owns 32 columns of C and keeps
//     its K x 32 slice of B as MFMA fragments, so the k-loop reads only A fragments from LDS and B is fetched once
//     per workgroup instead of once per tile;
//   * the next A tile travels global -> registers while the current one is multiplied; two LDS stages, one barrier
//     per tile; the BatchNorm + ReLU of the producing layer is applied while staging ("normalise on load");
//   * epilogue: bias, stores (two 128-byte row segments per instruction), and per-column sums / sums of squares for
//     the next BatchNorm accumulated in registers over ALL tiles of the workgroup: one statistics slab per workgroup.
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered), same C/D layout as gemm.hip.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int SBM = 64;   // rows of A per tile
constexpr int SPAD = 4;   // LDS row padding (floats): conflict-free ds_read_b128 fragments for K = 64, 96, 128

struct StreamArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    long long lda, ldb, ldc;
    const float *a_scale, *a_shift;  // [K] or NULL
    const float *bias;               // [N] or NULL
    float *stats;                    // [gridDim.x][2][N] or NULL
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// WN: waves along N (2 or 4; 256 threads = 4 waves, WM = 4 / WN waves along M); KG = K / 8;
// BKC: B is [N][K] (NT) else [K][N] (NN); AFF: prologue on A
template <int WN, int KG, bool BKC, bool AFF>
__global__ __launch_bounds__(256, (KG <= 8 ? 4 : (KG <= 12 ? 3 : 2))) void gemm_stream_kernel(const StreamArgs g)
{
    constexpr int K = KG * 8;
    constexpr int WM = 4 / WN;
    constexpr int TM = SBM / (32 * WM);       // 32-row accumulator tiles per wave
    constexpr int LD = K + SPAD;
    constexpr int NV = SBM * K / 4 / 256;     // float4 staged per thread
    static_assert(NV * 256 * 4 == SBM * K, "tile divides evenly");
    __shared__ __attribute__((aligned(16))) float s_a[2][SBM * LD];
    __shared__ __attribute__((aligned(16))) float s_aff[2][K];
    __shared__ float s_red[WM][2][32 * WN];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int col = 32 * wn + li;
    const bool col_ok = col < g.N;   // wave-uniform for N % 32 == 0

    if (AFF) {
        for (int t = threadIdx.x; t < K; t += 256) { s_aff[0][t] = g.a_scale[t]; s_aff[1][t] = g.a_shift[t]; }
    }
    // this wave's slice of B as fragments: lane (li, lh) of k-group q holds B[col][8q + 4lh + 0..3]
    float4 bf[KG];
#pragma unroll
    for (int q = 0; q < KG; ++q) {
        const int k0 = 8 * q + 4 * lh;
        if (!col_ok) bf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (BKC) bf[q] = ld4(g.B + (long long)col * g.ldb + k0);
        else bf[q] = make_float4(g.B[(long long)k0 * g.ldb + col], g.B[(long long)(k0 + 1) * g.ldb + col],
                                 g.B[(long long)(k0 + 2) * g.ldb + col], g.B[(long long)(k0 + 3) * g.ldb + col]);
    }
    const float bias = (g.bias && col_ok) ? g.bias[col] : 0.f;
    float csum = 0.f, csq = 0.f;

    const int tiles = (g.M + SBM - 1) / SBM;
    float4 st[NV];
    auto load_tile = [&](int tile) {
        const int m0 = tile * SBM;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + 256 * p;
            const int row = id / (K / 4), c4 = id - row * (K / 4);
            const int gr = m0 + row;
            st[p] = ld4(g.A + (long long)(gr < g.M ? gr : g.M - 1) * g.lda + 4 * c4);
        }
    };
    auto store_tile = [&](int tile, float *dst) {
        const int m0 = tile * SBM;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + 256 * p;
            const int row = id / (K / 4), c4 = id - row * (K / 4);
            float4 x = st[p];
            if (AFF) {
                const float4 s = *reinterpret_cast<const float4 *>(&s_aff[0][4 * c4]);
                const float4 t = *reinterpret_cast<const float4 *>(&s_aff[1][4 * c4]);
                x.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f); x.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
                x.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f); x.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
            }
            if (m0 + row >= g.M) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(dst + row * LD + 4 * c4) = x;
        }
    };

    int tile = blockIdx.x;
    if (tile < tiles) load_tile(tile);
    if (AFF) __syncthreads();  // s_aff visible before the first staging
    for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
        float *As = s_a[it & 1];
        store_tile(tile, As);
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < tiles) load_tile(next);

        f32x16 acc[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const float *ap = As + (wm * 32 * TM + li) * LD + 4 * lh;
#pragma unroll
        for (int q = 0; q < KG; ++q) {
            float4 fa[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const float4 *>(ap + a * 32 * LD + 8 * q);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, bf[q].x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, bf[q].y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, bf[q].z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, bf[q].w, acc[a], 0, 0, 0);
            }
        }
        // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        if (col_ok) {
            const int m0 = tile * SBM + wm * 32 * TM + 4 * lh;
            const bool full = tile * SBM + SBM <= g.M;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * a + (r & 3) + 8 * (r >> 2);
                    if (!full && row >= g.M) continue;
                    const float v = acc[a][r] + bias;
                    csum += v;
                    csq += v * v;
                    g.C[(long long)row * g.ldc + col] = v;
                }
        }
    }
    if (g.stats) {
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (lh == 0) { s_red[wm][0][32 * wn + li] = csum; s_red[wm][1][32 * wn + li] = csq; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * g.N; t += 256) {
            const int which = t / g.N, c = t - which * g.N;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += s_red[w][which][c];
            g.stats[((long long)blockIdx.x * 2 + which) * g.N + c] = s;
        }
    }
}

int stream_grid(int M, int K)
{
    const int tiles = (M + SBM - 1) / SBM;
    const int cap = 256 * (K <= 64 ? 4 : (K <= 96 ? 3 : 2));   // CUs x resident workgroups (LDS: 2 stages of 64 x (K+4) floats)
    return tiles < cap ? tiles : cap;
}

template <int WN, int KG, bool BKC>
void launch_aff(const StreamArgs &g, int grid, hipStream_t st)
{
    if (g.a_scale) hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, BKC, true>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_stream_kernel<WN, KG, BKC, false>), dim3(grid), dim3(256), 0, st, g);
}

template <int WN, bool BKC>
int launch_k(const StreamArgs &g, int grid, hipStream_t st)
{
    switch (g.K) {
        case 64: launch_aff<WN, 8, BKC>(g, grid, st); break;
        case 96: launch_aff<WN, 12, BKC>(g, grid, st); break;
        case 128: launch_aff<WN, 16, BKC>(g, grid, st); break;
        default: return PRIFIT_EINVAL;
    }
    return prifit_check_launch();
}

}  // namespace

extern "C" {

int prifit_gemm_stream_supported(int layout, int M, int N, int K)
{
    return (layout == 0 || layout == 1) && M >= 32768 && (N == 64 || N == 96 || N == 128) &&
           (K == 64 || K == 96 || K == 128);
}

int prifit_gemm_stream_slabs(int M, int K) { return stream_grid(M, K); }

int prifit_gemm_stream_f32(int layout, int M, int N, int K, const float *A, long long lda, const float *B,
                           long long ldb, float *C, long long ldc, const float *a_scale, const float *a_shift,
                           const float *bias, float *col_stats, void *stream)
{
    if (!A || !B || !C || !prifit_gemm_stream_supported(layout, M, N, K) || lda < K || ldc < N ||
        ((a_scale == nullptr) != (a_shift == nullptr)) || (lda & 3) || ((uintptr_t)A & 15) ||
        (layout == 0 && ((ldb & 3) || ((uintptr_t)B & 15) || ldb < K)) || (layout == 1 && ldb < N))
        return PRIFIT_EINVAL;
    StreamArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_scale = a_scale; g.a_shift = a_shift; g.bias = bias; g.stats = col_stats;
    const int grid = stream_grid(M, K);
    hipStream_t st = as_stream(stream);
    if (N == 64) return layout == 0 ? launch_k<2, true>(g, grid, st) : launch_k<2, false>(g, grid, st);
    return layout == 0 ? launch_k<4, true>(g, grid, st) : launch_k<4, false>(g, grid, st);
}

}  // extern "C"
