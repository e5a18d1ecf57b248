// Flash-style fused mean-shift iteration (src/mean_shift.py:61-82) for D = 128 on gfx950.
//
// One workgroup = 64 points ("queries") of one shape; it streams the whole dictionary X of that shape
// through LDS in tiles of 64 rows ("keys") and never materialises the N x N score matrix in LDS or
// registers beyond one 32 x 32 MFMA tile per wave:
//     S^T tile = X_sub . Z_q^T          (v_mfma_f32_32x32x2_f32, keys on the accumulator rows,
//                                        queries on the lanes)
//     P        = exp(clamp((S - 1) / b^2, -13, 75))      in registers
//     O_q     += P . X_sub              the accumulator registers of P are fed straight back as the
//                                        A operand of the second MFMA (k index = key = register index)
// Waves 0/1 own query rows 0-31 / 32-63 for the even 32-key sub-tiles, waves 2/3 the odd ones; the two
// partial (O, rowsum) pairs are added through LDS at the end, followed by the normalisation epilogue
// new = Z + (O/rowsum - Z); out = new / |new|.
//
// MODE 0 (forward) optionally streams P^T out (KT[key][query], coalesced: queries on lanes) for the
// backward pass.  MODE 1 (backward, dZ): the same data flow with P replaced by
//     gS^T = (X_sub . gO_q^T + g_rowsum_q) * K^T / b^2   where the clamp was inactive,
// accumulating dZ_q += gS . X_sub, no normalisation.
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int D = 128;        // embedding width (models/pointnet2_part_seg_msg.py:46: extra_conv_emb 128 -> 128)
constexpr int QB = 64;        // queries per workgroup
constexpr int KB = 64;        // keys per LDS tile
constexpr int LDSW = D + 4;   // padded row: conflict-free ds_read_b128 fragments (132 * i mod 64 distinct)

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Buffer addressing (SGPR resource + ONE constant VGPR offset per lane + a scalar offset per access) for every global
// access of the loop: the flat form costs two 64-bit VALU adds per access (16 stream stores, 8 tile loads and, in MODE 1,
// 16 stream loads per lane and step), and that address arithmetic -- not the memory system -- was what the streams cost:
// forward 530 -> 478 us with identical instructions otherwise (tools/micro/msf_variants.hip V0 -> V13).  The hardware
// bounds check of the resource (num_records = the exact extent) returns 0 for rows beyond N and drops stores beyond
// the matrix, so no access is predicated and nothing is selected after a load.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *p, long long bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
}
constexpr int OOB = 0x7fffffff;   // lane offset that fails the bounds check (invalid query column)

// global [rows x 128] tile -> registers (8 float4 per thread, 256 threads); rows beyond the matrix read as zeros
// (the scalar offset is not part of the bounds check: EXACT = every row of the tile exists, the row goes into the scalar
// offset; otherwise it goes into the lane offset, OOB for rows beyond the matrix)
template <int NTH = 256>
struct Tile64T {
    static constexpr int NV = 2048 / NTH, RP = NTH / 32;   // float4 per thread, rows covered by one pass of the block
    float4 v[NV];
    template <bool EXACT>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int voff, int row0, int nrows)
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) {   // thread's float4 p: row RP p + (tid >> 5), columns 4 (tid & 31) ..
            const int row = row0 + RP * p;
            const f32x4 t = __builtin_bit_cast(f32x4, EXACT
                ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff, row * D * 4, 0)
                : __builtin_amdgcn_raw_buffer_load_b128(rs, row + (int)(threadIdx.x >> 5) < nrows ? voff + row * D * 4 : OOB, 0, 0));
            v[p] = make_float4(t.x, t.y, t.z, t.w);
        }
    }
    template <bool EXACT>
    __device__ __forceinline__ void load_one(int p, __amdgpu_buffer_rsrc_t rs, int voff, int row0, int nrows)
    {
        const int row = row0 + RP * p;
        const f32x4 t = __builtin_bit_cast(f32x4, EXACT
            ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff, row * D * 4, 0)
            : __builtin_amdgcn_raw_buffer_load_b128(rs, row + (int)(threadIdx.x >> 5) < nrows ? voff + row * D * 4 : OOB, 0, 0));
        v[p] = make_float4(t.x, t.y, t.z, t.w);
    }
    __device__ __forceinline__ void store(float *__restrict__ lds) const
    {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int id = threadIdx.x + NTH * p;
            *reinterpret_cast<float4 *>(lds + (id >> 5) * LDSW + (id & 31) * 4) = v[p];
        }
    }
};
typedef Tile64T<256> Tile64;

// DIAGNOSIS build (PRIFIT_BUILD_DEFS=-DMSF_STAMPS, tools/msf_stamps.py): wave 0 of a few workgroups of the standard forward
// kernel leaves s_memtime stamps at the phase boundaries of its first key steps.  No stamp executes in the product build.
#ifdef MSF_STAMPS
__device__ unsigned long long g_msf_stamps[8 * 16 * 8];   // [workgroup slot][step][phase]
#define MSF_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (st_slot >= 0 && st_step < 16) {                                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
            if (threadIdx.x == 0) g_msf_stamps[(st_slot * 16 + st_step) * 8 + (i)] = t_;                    \
        }                                                                                                    \
    } while (0)
#else
#define MSF_STAMP(i) do { } while (0)
#endif

// the key tile of a step: ROWS = 64 or 128 rows loaded by NTH threads (one or two 64-row passes)
template <int NTH, int ROWS>
struct KeyTile {
    static constexpr int NT = ROWS / 64;
    Tile64T<NTH> t[NT];
    template <bool EXACT>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int voff, int row0, int nrows)
    {
#pragma unroll
        for (int i = 0; i < NT; ++i) t[i].template load<EXACT>(rs, voff, row0 + 64 * i, nrows);
    }
    __device__ __forceinline__ void store(float *__restrict__ lds) const
    {
#pragma unroll
        for (int i = 0; i < NT; ++i) t[i].store(lds + 64 * i * LDSW);
    }
};
}  // namespace

// Q: the query-side operand rows (Z for MODE 0, gO for MODE 1) [B,N,128]; X: dictionary [B,N,128].
// FAST: N % 64 == 0 -- every row / column of every tile exists, so the row part of each address is a SCALAR offset (no
// vector arithmetic per access at all); otherwise rows beyond the matrix are masked through the lane offset (OOB).
// MODE 0: the 16 stream values of a step are stored at the start of the NEXT step, ahead of that step's tile loads (vmcnt
// is one in-order counter for loads and stores on gfx9: the wait for a tile also waits for every older store, and a
// whole step later the stores have long retired).
//
// SK (stream-K, needs FAST): B x N/64 query blocks of N/64 key steps each are 768 workgroups of equal length at B = 24,
// N = 2048, for 512 resident slots: one and a half rounds.  A lone wave keeps its SIMD's matrix pipe ~55 % busy, two
// co-resident ones ~90 % (SQ counters: MFMA busy per wave cycle 0.50 at an average residency of 1.57 waves per SIMD), so
// the half-empty second round runs at half efficiency.  With SK the (query block, key step) units of the whole launch are
// cut into equal contiguous ranges for a grid of exactly the resident slots; a range is a sequence of SEGMENTS (part of
// one query block each).  A segment that covers all key steps of its block ends as before; a partial one adds its
// contribution with float atomics into a zero-initialised output (MODE 1: dZ; MODE 0: O and the row sums, the
// normalisation epilogue of those blocks runs in a second small launch).
//
// NW (4 or 8 waves): the 8-wave form keeps the 64 queries and steps through the keys 128 at a time (four key sub-tiles
// instead of two; the four partial (O, rowsum) pairs are added through LDS in two rounds).  It needs 102 KB of LDS, so it
// runs ONE workgroup per CU with the same 2 waves per SIMD as two 4-wave workgroups: the launcher uses it for the
// blocks that are left over after the full rounds of 4-wave workgroups (B x N / 64 = 768 blocks on 512 slots: the last
// 256 used to run one 4-wave workgroup per CU, a lone wave per SIMD, at ~55 % of the matrix pipe).
// NQG (2 or 1 query groups of 32): NW = 4, NQG = 1 is the "narrow" form -- 32 queries, the four waves take four key
// sub-tiles of a 128-key step, no query tile in LDS (MODE 0 keeps the query fragments in registers; the epilogue reads Z
// from global): 68 KB of LDS, so TWO independent workgroups per CU like the standard form.
// SLOAD (MODE 0, standard form, FAST): the FIRST update of a trajectory.  Z_0 = X (src/mean_shift.py:60), so its score matrix
// S = X X^T is symmetric and already in HBM: the bandwidth step has just written the chord matrix C = 2 - 2 X X^T
// (src/mean_shift.py:154-158), s = 1 - C / 2.  The S product -- half of the update's matrix work -- is replaced by four 16-byte
// loads per lane and step of C[query][key .. key + 3] (GST carries C; C is symmetric, and read along the QUERY's row a
// workgroup streams 64 contiguous 8 KB rows: read as C[key][query], 256-byte pieces of 2048 different rows, the same bytes took
// 303 instead of 293 us per launch against 394 for the standard kernel), requested one step ahead, behind the transform and ahead of the 64 MFMAs of O += P X; the
// query fragments are not needed at all.  fl(1 - fl(2 - 2 d) / 2)
// differs from the S product's own d by at most one rounding of (1 - d): the same order as the product's accumulation error.
template <int MODE, bool FAST, bool SK = false, int NW = 4, int NQG = 2, bool SLOAD = false>
__global__ __launch_bounds__(NW * 64, 2) void ms_fused_kernel(
    const float *__restrict__ Q, long long q_stride, const float *__restrict__ X, const float *__restrict__ bw,
    int N, const float *__restrict__ row_add,   // q_stride: batch stride of Q; row_add (MODE 1): g_rowsum [B,N]
    float *__restrict__ KT,              // MODE 0: out (may be NULL); MODE 1: in.  [B, N(keys), ldk] (queries contiguous)
    long long ldk, long long sk,         // row / batch stride of KT and GST
    float *__restrict__ GST,             // MODE 1: optional output gS^T, same layout as KT
    const float *__restrict__ Zin,       // MODE 0: current points (== Q) for the epilogue
    float *__restrict__ out,             // MODE 0: normalised new points; MODE 1: dZ   [B,N,128]
    float *__restrict__ O_out, float *__restrict__ rsum_out, float *__restrict__ nrm_out,  // MODE 0 saves
    int nbatch,                          // number of shapes
    int blk0,                            // !SK: linear index (shape-major) of this launch's first query block
    int xcd_map)                         // place whole shapes on one XCD (see below)
{
    static_assert((NW == 4 && NQG == 2) || (!SK && ((NW == 8 && NQG == 2) || (NW == 4 && NQG == 1 && MODE == 0))),
                  "standard, wide or (forward only) narrow form");
    static_assert(!SLOAD || (MODE == 0 && FAST && !SK && NW == 4 && NQG == 2), "SLOAD: the standard forward form on whole tiles");
    constexpr int QB = 32 * NQG;         // queries per workgroup (shadows the standard form's constant)
    constexpr int NKH = NW / NQG;        // key sub-tiles per step
    constexpr int KB = 32 * NKH;         // keys per step
    // (a double-buffered key tile with ONE barrier per step -- in the LDS the query tile leaves free -- measured SLOWER,
    // 419.7 against 408.5 us per launch, alternating builds on one box (228 instead of 209 VGPRs): not kept)
    constexpr bool SQ = NQG == 2;        // the query tile lives in LDS
    constexpr bool ILV = MODE == 0;      // output column of accumulator block d, lane li: 4 li + d instead of 32 d + li
    __shared__ __attribute__((aligned(16))) float s_q[SQ ? QB * LDSW : 4];
    __shared__ __attribute__((aligned(16))) float s_x[KB * LDSW];
    __shared__ float s_rs[NKH * QB];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int qg = wave % NQG, kh = wave / NQG;  // query group (32 rows), key sub-tile
    // segments of this workgroup: [u, uend) in units of (query block, key step)
    const int nsteps = (N + KB - 1) / KB, nqb = (N + QB - 1) / QB;
    int u = 0, uend = 1;
  for (bool first_seg = true; first_seg || (SK && u < uend); first_seg = false) {
    int b, q0, kbeg, kend;
    if (SK) {
        if (first_seg) {
            // XCD placement (consecutive workgroup ids are dealt round-robin over the 8 XCDs): the workgroups of one XCD
            // take CONSECUTIVE unit ranges, i.e. whole shapes -- a shape's dictionary X stays in one L2 and the rows of its
            // K^T / gS^T streams are read / written by neighbours on the same L2
            const int vid = (xcd_map && (gridDim.x & 7) == 0) ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
            const long long total = (long long)nbatch * nqb * nsteps;
            u = (int)(total * vid / gridDim.x);
            uend = (int)(total * (vid + 1) / gridDim.x);
            if (u >= uend) return;
        }
        const int blk = u / nsteps, s0 = u - blk * nsteps;
        const int s1 = min(nsteps, s0 + (uend - u));
        b = blk / nqb; q0 = (blk - b * nqb) * QB; kbeg = s0 * KB; kend = s1 * KB;
        u += s1 - s0;
    } else {
        const int L = blockIdx.y * gridDim.x + blockIdx.x + blk0;
        if (xcd_map && blk0 == 0 && (nbatch & 7) == 0) {   // same placement: XCD x works on the shapes x, x + 8, ...
            const int xcd = L & 7, j = L >> 3;
            b = (j / nqb) * 8 + xcd; q0 = (j % nqb) * QB;
        } else {
            b = L / nqb; q0 = (L - b * nqb) * QB;
        }
        kbeg = 0; kend = N;
    }
    const bool whole = !SK || (kend - kbeg >= N);
#ifdef MSF_STAMPS
    int st_slot = -1, st_step = 0;
    if (MODE == 0 && !SK && NW == 4 && NQG == 2 && threadIdx.x < 64) {
        const int Lb = blockIdx.y * gridDim.x + blockIdx.x;
        st_slot = (Lb == 3) ? 0 : (Lb == 130) ? 1 : (Lb == 301) ? 2 : (Lb == 470) ? 3 : (Lb == 515) ? 4 : (Lb == 600) ? 5 : (Lb == 700) ? 6 : (Lb == 767) ? 7 : -1;
    }
#endif
    const float *Qb = Q + (size_t)b * q_stride;
    const float *Xb = X + (size_t)b * N * D;
    const float bwv = bw[b];
    const float rcp_b2 = 1.0f / (bwv * bwv);
    const float c_e2 = rcp_b2 * 1.44269504088896341f;   // log2(e) / b^2
    const float kmin = __expf(-13.0f);

    const __amdgpu_buffer_rsrc_t q_rs = make_rsrc(Qb, (long long)N * D * 4), x_rs = make_rsrc(Xb, (long long)N * D * 4);
    const int t_voff = ((threadIdx.x >> 5) * D + (threadIdx.x & 31) * 4) * 4;
    KeyTile<NW * 64, KB> t;
    if (SQ) {
        Tile64T<NW * 64> tq;
        tq.template load<FAST>(q_rs, t_voff, q0, N);
        tq.store(s_q);
    }
    t.template load<FAST>(x_rs, t_voff, kbeg, N);

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float rsum = 0.f;
    const int qrow = qg * 32 + li;             // this lane's query row inside the tile
    const int gq = q0 + qrow;                  // global query index
    const bool q_ok = gq < N;
    const float radd = (MODE == 1 && q_ok) ? row_add[(size_t)b * N + gq] : 0.f;
    float *KTb = KT ? KT + (size_t)b * sk : nullptr;
    float *GSb = ((MODE == 1 || SLOAD) && GST) ? GST + (size_t)b * sk : nullptr;   // (SLOAD: the chord matrix, read only)
    // the N x N streams: element (key, query) at key * ldk + query; this lane's constant part = its query column and
    // the first key row of its accumulator registers, the scalar part = the step's key block and the register's row
    const long long nn_bytes = ((long long)(N - 1) * ldk + N) * 4;
    const __amdgpu_buffer_rsrc_t kt_rs = make_rsrc(KTb, KTb ? nn_bytes : 0), gs_rs = make_rsrc(GSb, GSb ? nn_bytes : 0);
    const int nn_voff = q_ok ? (int)(((long long)(kh * 32 + 4 * lh) * ldk + gq) * 4) : OOB;
    // (readfirstlane: the scalar offsets must BE scalar for the compiler -- with the row stride in a VGPR every buffer access
    // became a readfirstlane "waterfall" loop)
    const int ldk4 = __builtin_amdgcn_readfirstlane((int)ldk * 4);
    // (lane offset, scalar offset) of stream element (key block k + accumulator register r, this lane's query)
    auto st_voff = [&](int k, int r) {
        const int key = k + (r & 3) + 8 * (r >> 2);
        return FAST ? nn_voff : ((q_ok && key + kh * 32 + 4 * lh < N) ? nn_voff + key * ldk4 : OOB);
    };
    int row_off[16];   // loop-invariant scalar byte offsets of the 16 accumulator rows
#pragma unroll
    for (int r = 0; r < 16; ++r) row_off[r] = __builtin_amdgcn_readfirstlane(((r & 3) + 8 * (r >> 2)) * ldk4);
    auto st_soff = [&](int kbase_bytes, int r) { return FAST ? kbase_bytes + row_off[r] : 0; };
    // MODE 0: the query fragments (B operand of the S product) stay in registers for the whole kernel
    float4 qf[(MODE == 0 && !SLOAD) ? D / 8 : 1];
    if (MODE == 0 && !SLOAD) {
        const int q_voff = q_ok ? (gq * D + lh * 4) * 4 : OOB;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q_rs, q_voff, g * 32, 0));
            qf[g] = make_float4(v.x, v.y, v.z, v.w);
        }
    }

    float pprev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pprev[r] = 0.f;
    // SLOAD: the chord values under this wave's sub-tile of the NEXT step: accumulator registers 4 j .. 4 j + 3 of lane (query li,
    // half lh) are the keys k0 + 32 kh + 8 j + 4 lh + (0..3) -- one float4 of the query's row of C
    f32x4 cnext[SLOAD ? 4 : 1];
    const int c_voff = (SLOAD && q_ok) ? (int)(((long long)gq * ldk + kh * 32 + 4 * lh) * 4) : OOB;
    if (SLOAD) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cnext[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs_rs, c_voff, (kbeg + 8 * j) * 4, 0));
    }
    int it = 0;
    for (int k0 = kbeg; k0 < kend; k0 += KB, ++it) {
        float *sx = s_x;                 // this step's key tile
        MSF_STAMP(0);
        __syncthreads();                 // previous tile's readers are done (also orders the s_q store)
        MSF_STAMP(1);
        t.store(s_x);
        MSF_STAMP(2);
        __syncthreads();
        MSF_STAMP(3);
        const int kb_bytes = __builtin_amdgcn_readfirstlane(k0 * ldk4);          // this step's key block
        // MODE 0 (standard form): the step's 24 vector-memory instructions -- 8 loads of the next tile, 16 stores of the
        // previous step's K^T values -- are issued one per group of four S MFMAs below instead of in a block here: their
        // issue (~40 cycles each, in-kernel stamps: 970-1290 cycles per step with the matrix pipe idle) then runs beside
        // the matrix instructions.
        constexpr bool SPREAD = MODE == 0 && NW == 4 && NQG == 2 && !SLOAD;
        const bool st_prev = SPREAD && KTb && k0 > kbeg, ld_next = SPREAD && k0 + KB < kend;
        const int pb_bytes_s = __builtin_amdgcn_readfirstlane((k0 - KB) * ldk4);
        if (!SPREAD && (MODE == 0 ? KTb : GSb) && k0 > kbeg) {   // the previous step's stream values (K^T / gS^T)
            const int pb_bytes = __builtin_amdgcn_readfirstlane((k0 - KB) * ldk4);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pprev[r]), MODE == 0 ? kt_rs : gs_rs,
                                                      st_voff(k0 - KB, r), st_soff(pb_bytes, r), MODE == 0 ? 2 : 0);
        }
        if (!SPREAD && k0 + KB < kend) t.template load<FAST>(x_rs, t_voff, k0 + KB, N);

        // MODE 1: the saved kernel values under this wave's sub-tile, requested before the S MFMAs so that their
        // latency hides behind the 64 matrix instructions (they were the exposed part of this mode)
        float kfv[16];
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)   // rows / columns beyond the matrix come back as 0 (bounds check): K = 0 < kmin below
                kfv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(kt_rs, st_voff(k0, r), st_soff(kb_bytes, r), 0));
            // keep them HERE: left alone, the scheduler sinks the 16 loads behind the S MFMAs, right in front of their use
            // (fewer live registers), and every step then waits out the full memory latency
            __builtin_amdgcn_sched_barrier(0);
        }
        MSF_STAMP(4);
        // ---- S^T sub-tile (32 keys x 32 queries), K = 128: A = X_sub rows, B = query rows
        const float *xa = sx + (kh * 32 + li) * LDSW + lh * 4;
        const float *qb = s_q + qrow * LDSW + lh * 4;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = SLOAD ? fmaf(-0.5f, cnext[SLOAD ? r >> 2 : 0][r & 3], 1.0f) : 0.f;   // SLOAD: s = 1 - chord / 2
        constexpr int PFD = 1;                // groups requested ahead (ring of PFD + 1 fragments); 2 measured the same
        if constexpr (!SLOAD) {
        // the fragments of k group g + 1 are requested before the four MFMAs of group g (left to itself the compiler reads
        // each group right in front of its MFMAs and waits out the LDS latency with lgkmcnt(0), sixteen times per step)
        float4 a2[PFD + 1], b2[PFD + 1];
#pragma unroll
        for (int g = 0; g < PFD; ++g) {
            a2[g] = *reinterpret_cast<const float4 *>(xa + g * 8);
            if (MODE != 0) b2[g] = *reinterpret_cast<const float4 *>(qb + g * 8);
        }
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            if (g + PFD < D / 8) {
                a2[(g + PFD) % (PFD + 1)] = *reinterpret_cast<const float4 *>(xa + (g + PFD) * 8);
                if (MODE != 0) b2[(g + PFD) % (PFD + 1)] = *reinterpret_cast<const float4 *>(qb + (g + PFD) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float4 a = a2[g % (PFD + 1)];
            const float4 bq = MODE == 0 ? qf[g] : b2[g % (PFD + 1)];
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, sacc, 0, 0, 0);
            if (SPREAD) {
                if (ld_next && g < 8) t.t[0].template load_one<FAST>(g, x_rs, t_voff, k0 + KB, N);
                if (st_prev)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pprev[g]), kt_rs, st_voff(k0 - KB, g),
                                                          st_soff(pb_bytes_s, g), 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }   // !SLOAD
        // (MODE 1: nothing of the transform -- e.g. the compares on the just-requested K values -- may move above the S MFMAs)
        if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
#ifdef MSF_STAMPS
        asm volatile("" ::"v"(sacc[15]));   // the S product is complete in the register file before the stamp
#endif
        MSF_STAMP(5);
        // ---- elementwise transform; accumulator register r of lane (query li, half lh) is key (r&3)+8(r>>2)+4lh
        const int key_base = k0 + kh * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key_base + (r & 3) + 8 * (r >> 2);
            const bool ok = FAST || (key < N && q_ok);
            float p;
            if (MODE == 0) {
                // src/mean_shift.py:65-68 with src/guard.py:6-11: K = exp(clamp(-(2 - 2 s) / b^2 / 2, -13, 75)), evaluated as
                // exp2(clamp((s - 1) * log2(e) / b^2, -13 log2(e), 75 log2(e))): one fma, one med3, one v_exp_f32 per element
                // instead of seven VALU instructions -- on this chip every VALU instruction of an fp32-MFMA loop costs its
                // four issue cycles in full (tools/micro/mfma_shape_clock.hip: the matrix and the vector pipe do not overlap)
                const float t = fminf(fmaxf(fmaf(sacc[r], c_e2, -c_e2), -13.0f * 1.44269504088896341f), 75.0f * 1.44269504088896341f);
                p = ok ? __builtin_amdgcn_exp2f(t) : 0.f;
                rsum += p;
                pprev[r] = p;
            } else {
                const float kf = kfv[r];
                p = kf > kmin ? (sacc[r] + radd) * kf * rcp_b2 : 0.f;
                pprev[r] = p;
            }
            sacc[r] = p;
        }
#ifdef MSF_STAMPS
        asm volatile("" ::"v"(sacc[15]));
#endif
        if (SLOAD && k0 + KB < kend) {   // the next step's chord values: their latency runs under the 64 MFMAs below
#pragma unroll
            for (int j = 0; j < 4; ++j)
                cnext[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs_rs, c_voff, (k0 + KB + 8 * j) * 4, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
        MSF_STAMP(6);
        // ---- O_q += P . X_sub : A = P from the accumulator registers (k = key), B = X_sub[key][d]
        // Output column of accumulator block d, lane li: 4 li + d (NOT 32 d + li): the four B operands of a key row are then
        // four consecutive floats of the LDS row -- ONE ds_read_b128 per row instead of two ds_read2_b32 (in-kernel stamps:
        // the S product, fed by ds_read_b128, runs its 64 MFMAs in 4240 cycles, this one, fed by 32 ds_read2_b32, took
        // 5200-6400) -- and an output row of a lane is one 128-bit store.  The operands of key row r + 1 are requested before
        // the four MFMAs of row r (left to itself the compiler reads them right in front of their MFMAs with lgkmcnt(0)).
        // (ILV is the forward only: the dZ mode's split blocks add their accumulators with float atomics, and an atomic
        // instruction over columns 4 li + d touches four times as many cache lines as one over 32 d + li: 387 -> 408 us)
        if constexpr (ILV) {
            const float *xs = sx + (kh * 32 + 4 * lh) * LDSW + 4 * li;
            float4 bo[PFD + 1];
#pragma unroll
            for (int r = 0; r < PFD; ++r) bo[r] = *reinterpret_cast<const float4 *>(xs + ((r & 3) + 8 * (r >> 2)) * LDSW);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + PFD < 16)
                    bo[(r + PFD) % (PFD + 1)] = *reinterpret_cast<const float4 *>(xs + (((r + PFD) & 3) + 8 * ((r + PFD) >> 2)) * LDSW);
                __builtin_amdgcn_sched_barrier(0);   // (the read stays ahead of the MFMAs it does not feed)
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], bo[r % (PFD + 1)].x, oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], bo[r % (PFD + 1)].y, oacc[1], 0, 0, 0);
                oacc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], bo[r % (PFD + 1)].z, oacc[2], 0, 0, 0);
                oacc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], bo[r % (PFD + 1)].w, oacc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const float *xs = sx + (kh * 32 + 4 * lh) * LDSW + li;
            float bo[2][4];
#pragma unroll
            for (int d = 0; d < 4; ++d) bo[0][d] = xs[32 * d];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
                    const float *row = xs + (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * LDSW;
#pragma unroll
                    for (int d = 0; d < 4; ++d) bo[(r + 1) & 1][d] = row[32 * d];
                }
                __builtin_amdgcn_sched_barrier(0);   // (the reads stay ahead of the MFMAs they do not feed)
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[r], bo[r & 1][d], oacc[d], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef MSF_STAMPS
        asm volatile("" ::"v"(oacc[3][15]));   // the O product is complete
        MSF_STAMP(7);
        ++st_step;
#endif
    }

    if (MODE == 0 ? KTb != nullptr : GSb != nullptr) {  // the last step's stream values
        const int kl = ((kend + KB - 1) / KB - 1) * KB;
        const int kl_bytes = __builtin_amdgcn_readfirstlane(kl * ldk4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pprev[r]), MODE == 0 ? kt_rs : gs_rs,
                                                  st_voff(kl, r), st_soff(kl_bytes, r), MODE == 0 ? 2 : 0);
    }
    // ---- combine the key sub-tiles: a binary tree through LDS (the X tile is dead now); one slot = a partial O tile of both
    // query groups [64 queries][128], 32 KiB; NW = 4: one round (waves 2, 3 park, waves 0, 1 add), NW = 8: two rounds
    __syncthreads();
    float *s_part = s_x;  // NKH / 2 slots of 32 KiB <= KB * LDSW floats
    if (MODE == 0) {
        rsum += __shfl_xor(rsum, 32, 64);  // both key sub-rows of this wave
        if (lh == 0) s_rs[kh * QB + qrow] = rsum;
    }
#pragma unroll
    for (int step = 1; step < NKH; step <<= 1) {
        float *slot = s_part + (kh / (2 * step)) * (QB * D);
        if ((kh & (2 * step - 1)) == step) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;  // O accumulator: row = query, col (lane) = d
                    slot[(qg * 32 + qr) * D + 32 * d + li] = oacc[d][r];
                }
        }
        __syncthreads();
        if ((kh & (2 * step - 1)) == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    oacc[d][r] += slot[(qg * 32 + qr) * D + 32 * d + li];
                }
        }
        if (2 * step < NKH) __syncthreads();   // the slot is parked into again
    }
    if (kh == 0) {

    float *outb = out + (size_t)b * N * D;
    if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = q0 + qg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (gr >= N) continue;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (whole) outb[(size_t)gr * D + 32 * d + li] = oacc[d][r];
                else unsafeAtomicAdd(outb + (size_t)gr * D + 32 * d + li, oacc[d][r]);   // dZ is zero-initialised (SK)
            }
        }
    } else if (!whole) {
        // MODE 0, partial segment: O and the row sums accumulate in the (zero-initialised) saved buffers; the
        // normalisation of these blocks is prifit_meanshift_update_fwd's job (second launch)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = qg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int gr = q0 + qr;
#pragma unroll
            for (int d = 0; d < 4; ++d) unsafeAtomicAdd(O_out + ((size_t)b * N + gr) * D + 4 * li + d, oacc[d][r]);
            if (li == 0) unsafeAtomicAdd(rsum_out + (size_t)b * N + gr, s_rs[qr] + s_rs[QB + qr]);
        }
    } else {
    // ---- MODE 0 epilogue (src/mean_shift.py:70-82): Mv = O / rowsum; new = Z + (Mv - Z); out = new / |new|
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = qg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int gr = q0 + qr;
        float rs = s_rs[qr];
#pragma unroll
        for (int k = 1; k < NKH; ++k) rs += s_rs[k * QB + qr];
        const float dinv = 1.0f / rs;
        float nv[4];
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float z = SQ ? s_q[qr * LDSW + 4 * li + d] : (gr < N ? Zin[((size_t)b * N + gr) * D + 4 * li + d] : 0.f);
            const float m = oacc[d][r] * dinv - z;
            nv[d] = z + m;
            ss += nv[d] * nv[d];
        }
        // sum over the 32 lanes of this half (the lane's four columns already added locally)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float nrm = sqrtf(ss);
        if (gr < N) {
            *reinterpret_cast<float4 *>(outb + (size_t)gr * D + 4 * li) = make_float4(nv[0] / nrm, nv[1] / nrm, nv[2] / nrm, nv[3] / nrm);
            if (O_out)
                *reinterpret_cast<float4 *>(O_out + ((size_t)b * N + gr) * D + 4 * li) =
                    make_float4(oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]);
            if (li == 0) {
                nrm_out[(size_t)b * N + gr] = nrm;
                rsum_out[(size_t)b * N + gr] = rs;
            }
        }
    }
    }
    }   // kh == 0
    if (SK) __syncthreads();   // the next segment overwrites s_q / s_x
  }     // segments
}

// Backward w.r.t. the dictionary X (both of its uses in one iteration), key-major:
//     dX_j += sum_q gS[q][j] Z_q + sum_q K[q][j] gO_q,   gS = (gO X^T + g_rowsum 1^T) * K / b^2 (clamp-masked)
// One workgroup = 64 keys; it streams the query-side rows (gO and Z, 64 per tile) through LDS.  Waves 0/1
// own keys 0-31 / 32-63 for query sub-tile 0, waves 2/3 for sub-tile 1 (combined through LDS at the end).
//     T tile = gO_sub . X_j^T           (queries on the accumulator rows, keys on the lanes; X_j fragments
//                                        stay in registers for the whole kernel)
//     gS, K                             in registers (K^T is read as 4 float4 per lane along the query axis)
//     dX_j  += gS^T . Z_sub + K^T . gO_sub   both A operands come straight from registers (k index = query)
__global__ __launch_bounds__(256, 2) void ms_fused_dx_kernel(
    const float *__restrict__ gO, const float *__restrict__ Zc, const float *__restrict__ X,
    const float *__restrict__ bw, const float *__restrict__ row_add, const float *__restrict__ KT, long long ldk,
    long long sk, int N, float *__restrict__ dX)
{
    __shared__ __attribute__((aligned(16))) float s_g[QB * LDSW];   // gO tile (64 queries)
    __shared__ __attribute__((aligned(16))) float s_z[QB * LDSW];   // Z tile
    __shared__ float s_ra[QB];

    const int b = blockIdx.y, k0 = blockIdx.x * KB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int kg = wave & 1, qh = wave >> 1;  // key group (32 keys), query half of every 64-query tile
    const float *Gb = gO + (size_t)b * N * D;
    const float *Zb = Zc + (size_t)b * N * D;
    const float *Xb = X + (size_t)b * N * D;
    const float *KTb = KT + (size_t)b * sk;
    const float *RAb = row_add + (size_t)b * N;
    const float bwv = bw[b];
    const float rcp_b2 = 1.0f / (bwv * bwv);
    const float kmin = __expf(-13.0f);

    // this lane's key and its X row fragments (B operand of T = gO X^T): k = 8g + 4 lh + j
    const int gkey = k0 + kg * 32 + li;
    const bool key_ok = gkey < N;
    float4 xf[D / 8];
#pragma unroll
    for (int g = 0; g < D / 8; ++g) {
        xf[g] = ld4(Xb + (size_t)(key_ok ? gkey : 0) * D + g * 8 + lh * 4);
        if (!key_ok) xf[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    f32x16 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(Gb, (long long)N * D * 4), z_rs = make_rsrc(Zb, (long long)N * D * 4);
    const int t_voff = ((threadIdx.x >> 5) * D + (threadIdx.x & 31) * 4) * 4;
    Tile64 tg, tz;
    tg.load<false>(g_rs, t_voff, 0, N);
    tz.load<false>(z_rs, t_voff, 0, N);
    for (int q0 = 0; q0 < N; q0 += QB) {
        __syncthreads();
        tg.store(s_g);
        tz.store(s_z);
        if (threadIdx.x < QB) s_ra[threadIdx.x] = (q0 + threadIdx.x) < N ? RAb[q0 + threadIdx.x] : 0.f;
        __syncthreads();
        if (q0 + QB < N) { tg.load<false>(g_rs, t_voff, q0 + QB, N); tz.load<false>(z_rs, t_voff, q0 + QB, N); }

        // K values for (query register r, key lane): K^T[key][q0 + qh*32 + 8g + 4lh .. +3], 4 float4 per lane
        const int qbase = q0 + qh * 32 + 4 * lh;
        float4 kq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool ok = key_ok && (qbase + 8 * g + 3) < N;  // N % 4 == 0 is checked by the launcher
            kq[g] = ld4(KTb + (size_t)(key_ok ? gkey : 0) * ldk + (ok ? qbase + 8 * g : 0));
            if (!ok) kq[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // T[q][key] : A = gO_sub rows (query = lane), B = X_j fragments
        const float *ga = s_g + (qh * 32 + li) * LDSW + lh * 4;
        f32x16 tacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) tacc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4 *>(ga + g * 8);
            tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xf[g].x, tacc, 0, 0, 0);
            tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xf[g].y, tacc, 0, 0, 0);
            tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xf[g].z, tacc, 0, 0, 0);
            tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xf[g].w, tacc, 0, 0, 0);
        }
        // register r <-> query qh*32 + (r&3) + 8(r>>2) + 4lh ; kq[r>>2] component (r&3)
        float kf[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) { kf[4 * g] = kq[g].x; kf[4 * g + 1] = kq[g].y; kf[4 * g + 2] = kq[g].z; kf[4 * g + 3] = kq[g].w; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = qh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            tacc[r] = kf[r] > kmin ? (tacc[r] + s_ra[qr]) * kf[r] * rcp_b2 : 0.f;
        }
        // dX_j += gS^T . Z_sub + K^T . gO_sub   (A from registers: k = query; B rows of the LDS tiles)
        const float *zs = s_z + (qh * 32 + 4 * lh) * LDSW + li;
        const float *gs = s_g + (qh * 32 + 4 * lh) * LDSW + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * LDSW;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(tacc[r], zs[ro + 32 * d], acc[d], 0, 0, 0);
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[r], gs[ro + 32 * d], acc[d], 0, 0, 0);
            }
        }
    }
    // combine the two query halves and accumulate into dX
    __syncthreads();
    float *s_part = s_g;
    if (qh == 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s_part[(kg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + 32 * d + li] = acc[d][r];
    }
    __syncthreads();
    if (qh == 1) return;
    float *dXb = dX + (size_t)b * N * D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kr = kg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;  // accumulator row = key
        if (k0 + kr >= N) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float *dst = dXb + (size_t)(k0 + kr) * D + 32 * d + li;
            *dst += acc[d][r] + s_part[kr * D + 32 * d + li];
        }
    }
}

// dX += gS^T Z + K^T gO from the two N x N streams the other kernels wrote (gS^T by MODE 1, K^T by MODE 0), key-major,
// N % 64 == 0.  Both products sum over the QUERY index, and a 32x32x2 MFMA takes its A operand as one value per lane
// (lane = key row, lane half = k): the N x N operands therefore never pass through LDS -- every lane reads its own key
// row of gS^T / K^T straight from global memory, four 16-byte pieces per stream and step (k order 8g + 4h + j, the same
// the B rows are read in), prefetched one step ahead; only the small query-side tiles (Z and gO, 64 x 128 each) are
// staged in LDS and shared by the four waves.  This is the "PV phase" of the forward kernel twice -- no S product, no
// exp, no N x N store, no k-tile barrier per 32 deep slice as in the tiled GEMM, and no split-K atomics: a workgroup owns
// its 64 keys, so dX is bit-reproducible from run to run.  Measured 456 us per call against ~440 us for the dual-source
// GEMM at B = 24, N = 2048 (113 vs 117 TFLOP/s): opt-in (PRIFIT_MS_DX_STREAMS=1), the GEMM stays the default.
template <int NKG>   // key groups of 32 per workgroup (2: 64 keys, 256 threads; 4: 128 keys, 512 threads)
__global__ __launch_bounds__(NKG * 128, 2) void ms_dx_streams_kernel(
    const float *__restrict__ gO, const float *__restrict__ Zc, const float *__restrict__ GST, const float *__restrict__ KT,
    long long ldk, long long sk, int N, float *__restrict__ dX)
{
    __shared__ __attribute__((aligned(16))) float s_gz[2 * QB * LDSW];   // gO tile, Z tile (64 queries each)
    float *s_g = s_gz, *s_z = s_gz + QB * LDSW;

    const int b = blockIdx.y, k0 = blockIdx.x * (32 * NKG);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int kg = wave % NKG, qh = wave / NKG;  // key group (32 keys), query half of every 64-query tile
    const float *Gb = gO + (size_t)b * N * D;
    const float *Zb = Zc + (size_t)b * N * D;
    const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(Gb, (long long)N * D * 4), z_rs = make_rsrc(Zb, (long long)N * D * 4);
    const long long nn_bytes = ((long long)(N - 1) * ldk + N) * 4;
    const __amdgpu_buffer_rsrc_t gs_rs = make_rsrc(GST + (size_t)b * sk, nn_bytes), kt_rs = make_rsrc(KT + (size_t)b * sk, nn_bytes);
    const int t_voff = ((threadIdx.x >> 5) * D + (threadIdx.x & 31) * 4) * 4;
    // this lane's key row, first query of its half and lane half: element (key, q0 + qh*32 + 8g + 4lh + j)
    const int a_voff = (int)(((long long)(k0 + kg * 32 + li) * ldk + qh * 32 + 4 * lh) * 4);

    f32x16 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    Tile64T<NKG * 128> tg, tz;
    tg.template load<true>(g_rs, t_voff, 0, N);
    tz.template load<true>(z_rs, t_voff, 0, N);
    f32x4 an[8];   // next step's A fragments: [0..3] gS^T, [4..7] K^T
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        an[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs_rs, a_voff, g * 32, 0));
        an[4 + g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kt_rs, a_voff, g * 32, 0));
    }
    for (int q0 = 0; q0 < N; q0 += QB) {
        __syncthreads();
        tg.store(s_g);
        tz.store(s_z);
        f32x4 ac[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ac[i] = an[i];
        __syncthreads();
        if (q0 + QB < N) {
            tg.template load<true>(g_rs, t_voff, q0 + QB, N);
            tz.template load<true>(z_rs, t_voff, q0 + QB, N);
            const int sb = __builtin_amdgcn_readfirstlane((q0 + QB) * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                an[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs_rs, a_voff, sb + g * 32, 0));
                an[4 + g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kt_rs, a_voff, sb + g * 32, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // the prefetches stay ahead of this step's MFMAs
        // dX_j += gS^T . Z_sub + K^T . gO_sub   (A from registers: k = query 8g + 4lh + j; B rows of the LDS tiles)
        const float *zs = s_z + (qh * 32 + 4 * lh) * LDSW + li;
        const float *gs = s_g + (qh * 32 + 4 * lh) * LDSW + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * LDSW;
            const float a1 = ac[r >> 2][r & 3], a2 = ac[4 + (r >> 2)][r & 3];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, zs[ro + 32 * d], acc[d], 0, 0, 0);
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, gs[ro + 32 * d], acc[d], 0, 0, 0);
            }
        }
    }
    // combine the two query halves and accumulate into dX (the partials of NKG x 32 keys x 128 floats fit the two tiles)
    __syncthreads();
    float *s_part = s_g;   // s_g and s_z are adjacent: 2 * 64 * 132 floats >= NKG * 32 * 128
    static_assert(NKG * 32 * D <= 2 * QB * LDSW, "partials fit the tile buffers");
    if (qh == 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s_part[(kg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + 32 * d + li] = acc[d][r];
    }
    __syncthreads();
    if (qh == 1) return;
    float *dXb = dX + (size_t)b * N * D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kr = kg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;  // accumulator row = key
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float *dst = dXb + (size_t)(k0 + kr) * D + 32 * d + li;
            *dst += acc[d][r] + s_part[kr * D + 32 * d + li];
        }
    }
}

// whole shapes placed on one XCD (the dictionary in one L2, the rows of the K^T / gS^T streams written by neighbours on
// the same L2); measured 436 against 437 us without it: kept on, not a switch
static int xcd_map() { return 1; }

static int sk_slots()
{
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 512;
        return 2 * p.multiProcessorCount;   // two workgroups of the fused kernels per CU (68 KB of LDS, <= 256 VGPRs)
    }();
    return n;
}

extern "C" {

#ifdef MSF_STAMPS
int prifit_debug_msf_stamps(unsigned long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_msf_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

int prifit_meanshift_fused_fwd(const float *Z, const float *X, const float *bw, int B, int N, int D_, float *KT,
                               long long ld_kt, long long stride_kt, float *Znext, float *O, float *rowsum,
                               float *nrm, void *stream)
{
    if (!Z || !X || !bw || !Znext || !rowsum || !nrm || B <= 0 || N <= 0 || D_ != D || B > 65535 ||
        (KT && ld_kt < N))
        return PRIFIT_EINVAL;
    // (Left-over blocks after the full rounds -- B x N / 64 = 768 blocks on 512 slots -- run as they are: the forward's time
    // is linear in the number of blocks (B = 16 / 24 / 32: 345 / 504 / 655 us), it has no tail to win back; an 8-wave and a
    // 32-query "narrow" form for the last 256 blocks both lost 3-4 % to their second launch, DESIGN 5c, and are gone.)
    if (N % QB == 0)
        hipLaunchKernelGGL((ms_fused_kernel<0, true>), dim3(N / QB, B), dim3(256), 0, as_stream(stream), Z,
                           (long long)N * D, X, bw, N, (const float *)nullptr, KT, ld_kt, stride_kt, (float *)nullptr, Z,
                           Znext, O, rowsum, nrm, B, 0, xcd_map());
    else
        hipLaunchKernelGGL((ms_fused_kernel<0, false>), dim3((N + QB - 1) / QB, B), dim3(256), 0, as_stream(stream), Z,
                           (long long)N * D, X, bw, N, (const float *)nullptr, KT, ld_kt, stride_kt, (float *)nullptr, Z,
                           Znext, O, rowsum, nrm, B, 0, xcd_map());
    return prifit_check_launch();
}

int prifit_meanshift_fused_first_supported(int N, int D_) { return (N > 0 && N % QB == 0 && D_ == D) ? 1 : 0; }

int prifit_meanshift_fused_first_fwd(const float *X, const float *chord, long long ld_c, long long stride_c, const float *bw, int B,
                                     int N, int D_, float *Znext, float *O, float *rowsum, float *nrm, void *stream)
{
    if (!X || !chord || !bw || !Znext || !rowsum || !nrm || B <= 0 || B > 65535 || !prifit_meanshift_fused_first_supported(N, D_) ||
        ld_c < N || stride_c < (long long)(N - 1) * ld_c + N)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL((ms_fused_kernel<0, true, false, 4, 2, true>), dim3(N / QB, B), dim3(256), 0, as_stream(stream), X,
                       (long long)N * D, X, bw, N, (const float *)nullptr, (float *)nullptr, ld_c, stride_c,
                       const_cast<float *>(chord), X, Znext, O, rowsum, nrm, B, 0, xcd_map());
    return prifit_check_launch();
}

int prifit_meanshift_fused_bwd_dz(const float *gO, long long gO_batch_stride, const float *X, const float *bw,
                                  const float *g_rowsum, const float *KT, long long ld_kt, long long stride_kt,
                                  float *gST, int B, int N, int D_, float *dZ, int balanced, void *stream)
{
    if (!gO || !X || !bw || !g_rowsum || !KT || !dZ || B <= 0 || N <= 0 || D_ != D || B > 65535 || ld_kt < N ||
        gO_batch_stride < (long long)N * D)
        return PRIFIT_EINVAL;
    // stream-K: a grid of exactly the resident slots (2 workgroups per CU); dZ must then arrive zero-initialised
    const int slots = sk_slots();
    if (N % QB == 0 && balanced && (long long)B * (N / QB) > slots && ((long long)B * (N / QB)) % slots != 0)
        hipLaunchKernelGGL((ms_fused_kernel<1, true, true>), dim3(slots), dim3(256), 0, as_stream(stream), gO,
                           gO_batch_stride, X, bw, N, g_rowsum, const_cast<float *>(KT), ld_kt, stride_kt, gST,
                           (const float *)nullptr, dZ, (float *)nullptr, (float *)nullptr, (float *)nullptr, B, 0, xcd_map());
    else if (N % QB == 0)
        hipLaunchKernelGGL((ms_fused_kernel<1, true>), dim3(N / QB, B), dim3(256), 0, as_stream(stream), gO,
                           gO_batch_stride, X, bw, N, g_rowsum, const_cast<float *>(KT), ld_kt, stride_kt, gST,
                           (const float *)nullptr, dZ, (float *)nullptr, (float *)nullptr, (float *)nullptr, B, 0, xcd_map());
    else
        hipLaunchKernelGGL((ms_fused_kernel<1, false>), dim3((N + QB - 1) / QB, B), dim3(256), 0, as_stream(stream), gO,
                           gO_batch_stride, X, bw, N, g_rowsum, const_cast<float *>(KT), ld_kt, stride_kt, gST,
                           (const float *)nullptr, dZ, (float *)nullptr, (float *)nullptr, (float *)nullptr, B, 0, xcd_map());
    return prifit_check_launch();
}

int prifit_meanshift_dx_streams(const float *gO, const float *Z, const float *gST, const float *KT, long long ld_kt,
                                long long stride_kt, int B, int N, int D_, float *dX, void *stream)
{
    if (!gO || !Z || !gST || !KT || !dX || B <= 0 || N <= 0 || (N % QB) || (ld_kt & 3) || ld_kt < N || D_ != D || B > 65535 ||
        (((uintptr_t)gST | (uintptr_t)KT) & 15) || (stride_kt & 3))
        return PRIFIT_EINVAL;
    // (128 keys per 512-thread workgroup measured slower: 566 vs 456 us)
    hipLaunchKernelGGL(ms_dx_streams_kernel<2>, dim3(N / KB, B), dim3(256), 0, as_stream(stream), gO, Z, gST, KT, ld_kt,
                       stride_kt, N, dX);
    return prifit_check_launch();
}

int prifit_meanshift_fused_bwd_dx(const float *gO, const float *Z, const float *X, const float *bw,
                                  const float *g_rowsum, const float *KT, long long ld_kt, long long stride_kt,
                                  int B, int N, int D_, float *dX, void *stream)
{
    if (!gO || !Z || !X || !bw || !g_rowsum || !KT || !dX || B <= 0 || N <= 0 || (N & 3) || (ld_kt & 3) ||
        ld_kt < N || D_ != D || B > 65535)
        return PRIFIT_EINVAL;
    hipLaunchKernelGGL(ms_fused_dx_kernel, dim3((N + KB - 1) / KB, B), dim3(256), 0, as_stream(stream), gO, Z, X, bw,
                       g_rowsum, KT, ld_kt, stride_kt, N, dX);
    return prifit_check_launch();
}

}  // extern "C"
