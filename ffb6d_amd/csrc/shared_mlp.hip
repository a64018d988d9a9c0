// ffb6d_amd/csrc/shared_mlp.hip -- the "shared MLP" of FFB6D / RandLA-Net as one fused fp32 MFMA
// GEMM for gfx950.
//
// Reference: every 1x1 Conv + BatchNorm + activation wrapper on the path
//   ffb6d/models/pytorch_utils.py:75-129        (pt_utils.Conv1d/Conv2d: conv -> BN(eps 1e-5) -> ReLU)
//   ffb6d/models/RandLA/pytorch_utils.py:35-111 (conv -> BN(eps 1e-6) -> LeakyReLU(0.2))
// and the tensor plumbing around them in FFB6D.forward (ffb6d.py:245-263,273-298,302-307):
//   torch.cat((a, b), dim=1) -> conv            two K-ranges read from two tensors, no cat
//   conv(cat(a, interp(b)))                     W_a*a + gather(W_b*b): the interpolated half is a
//                                               column gather of a small pre-multiplied matrix
//   leaky(mlp2(f) + shortcut(x))  (RandLANet.py:179-184)   one GEMM over K = [f ; x]
//
//   out[b, m, p] = act( sum_k Wt[k, m] * X[b, k, p]  + bias[m]  + Y[b, m, gidx[b, p]] )
//
// with X = [X1 ; X2] stacked along k.  Channel-major activations (the reference's layout) make
// X a row-major K x P matrix per frame, so B-operand tiles are plain coalesced float4 row loads;
// weights arrive pre-transposed ([K, Cout]) with eval-mode BatchNorm folded in.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles, same peak as the fp32 vector rate, 157 TF;
// there is no TF32/xf32 on gfx950).  Operand layout: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; accumulator element r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
// Block = 256 threads = 4 waves, tile BM x 128 x 16; LDS tiles are k-major so a fragment read is
// 32 consecutive floats per half-wave (conflict free: lanes l and l+32 may share a bank).
#include <cstdlib>

#include "common.h"
#include "ffb6d_ops.h"

namespace ffb6d {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BLK = 256;
constexpr int BN = 128;
constexpr int BK = 16;

struct MlpParams {
    const float* wt;      // [K1+K2, Cout]
    const float* bias;    // [Cout] or null
    const float* x1;      // [B, K1, P]
    const float* x2;      // [B, K2, P] or null
    const float* yg;      // [B, Cout, Py] or null
    const void* gidx;     // [B, P] int32/int64 or null
    float* out;           // [B, Cout, P]
    long long x1_bs, x2_bs, yg_bs, out_bs;
    int k1, k2, cout, P, py, act, idx64;
    // flat mode (small P): tile columns run over all frames, c = b*P + p; blockIdx.z = K split
    int nb, kchunk;       // frames; k range per split (multiple of BK)
    int col_tiles, nz;    // grid decode: blockIdx.x = col_tile + col_tiles * z, z = frame (or K split when flat)
    float* part;          // [nsplit, cout, nb*P] partial sums when nsplit > 1
};

__device__ __forceinline__ float activate(float v, int act)
{
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : 0.2f * v;
    return v;
}

// WM x WN waves; each wave owns TM x TN MFMA tiles of 32x32
// all-reduce inside a DPP row of 16 lanes with row rotations (row_ror:8,4,2,1): VALU-rate, no
// LDS-crossbar traffic; every lane ends up with the reduction over its row
template <int ROR>
__device__ __forceinline__ float row_ror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + ROR, 0xf, 0xf, false));
}
// max over a DPP row: v_max_f32 with the rotated operand folded in (the compiler folds row_ror into v_add_f32 for
// the sums below, but for fmaxf it emits mov_dpp + two canonicalising v_max per step -- 22 instructions instead
// of 8).  s_nop 1: a DPP operand written by the previous VALU instruction needs two wait states, and the
// hazard recogniser does not look inside inline assembly.
__device__ __forceinline__ float row16_max(float v)
{
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

// ATT: attentive-pooling epilogue (RandLANet.py:243-248).  The GEMM is the score matrix
// A = W_fc * S over the feature set S = [x1 ; x2] ([d, N*16] per frame, 16 neighbours of a point in
// 16 consecutive columns); instead of storing A the epilogue forms softmax over each 16-column
// group, multiplies with S's own row m and writes the pooled [d, N] tensor.  A 16-column group is
// one 16-lane DPP row of the accumulator layout, so both reductions are 4 row shuffles.
template <int BM, bool FLAT, bool ATT = false>
__global__ void __launch_bounds__(BLK)
shared_mlp_kernel(const MlpParams p)
{
    constexpr int WM = BM == 128 ? 2 : 1;           // waves along M
    constexpr int WN = 4 / WM;                       // waves along N
    constexpr int TM = BM / (32 * WM);               // 32-row tiles per wave (128: 2, 64: 2, 32: 1)
    constexpr int TN = BN / (32 * WN);               // 32-col tiles per wave (128: 2, 64: 1, 32: 1)
    // two LDS stages: tile k+1 is written while tile k is multiplied -> one barrier per k-step
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: hardware sends workgroup i to XCD i % 8 (MI355X: 8 XCDs, private L2s).
    // gridDim.x is padded to a multiple of 8 and enumerates (column tile, frame / K-split) pairs,
    // gridDim.y the row tiles, so all row tiles that read the SAME activation columns carry the
    // same blockIdx.x and therefore share one XCD's L2 (the weight tiles are small and replicated).
    const int ncol_tiles = p.col_tiles;
    if ((int)blockIdx.x >= ncol_tiles * p.nz) return;
    const int bz = blockIdx.x / ncol_tiles;
    const int b = FLAT ? 0 : bz;
    const int m0 = blockIdx.y * BM;
    const int p0 = (blockIdx.x - bz * ncol_tiles) * BN;
    const int K = p.k1 + p.k2;
    const int kbeg = FLAT ? bz * p.kchunk : 0;
    const int kend = FLAT ? min(K, kbeg + p.kchunk) : K;
    const int ncols = FLAT ? p.nb * p.P : p.P;          // columns of this launch's N dimension
    const float* x1 = p.x1 + (size_t)b * p.x1_bs;
    const float* x2 = p.x2 ? p.x2 + (size_t)b * p.x2_bs : nullptr;
    // 16-byte loads only when every row start is 16-byte aligned
    const bool vecP = (p.P & 3) == 0 && ((p.x1_bs | p.x2_bs) & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) == 0;
    const bool vecM = (p.cout & 3) == 0 && (reinterpret_cast<uintptr_t>(p.wt) & 15) == 0;

    // global -> register staging: A tile BK x BM (rows = k, contiguous in m), B tile BK x BN
    constexpr int A_F4 = BK * BM / 4 / BLK;          // float4 per thread (128: 2, 64: 1, 32: 0.5 -> handled)
    constexpr int B_F4 = BK * BN / 4 / BLK;          // 2
    float4 ra[A_F4 > 0 ? A_F4 : 1], rb[B_F4];

    auto load_tiles = [&](int k0) {
        // A
        constexpr int A_TOTAL = BK * BM / 4;
#pragma unroll
        for (int i = 0; i < (A_F4 > 0 ? A_F4 : 1); ++i) {
            const int f = tid + i * BLK;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < A_TOTAL) {
                const int kr = f / (BM / 4), mc = (f % (BM / 4)) * 4;
                const int k = k0 + kr, m = m0 + mc;
                if (k < kend) {
                    const float* src = p.wt + (size_t)k * p.cout + m;
                    if (vecM && m + 3 < p.cout) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (m < p.cout) v.x = src[0];
                        if (m + 1 < p.cout) v.y = src[1];
                        if (m + 2 < p.cout) v.z = src[2];
                        if (m + 3 < p.cout) v.w = src[3];
                    }
                }
            }
            ra[i] = v;
        }
        // B
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * BLK;
            const int kr = f / (BN / 4), nc = (f % (BN / 4)) * 4;
            const int k = k0 + kr;
            int pp = p0 + nc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) {
                const float* row;
                if constexpr (FLAT) {   // column -> (frame, p); P % 4 == 0 keeps a float4 inside one frame
                    const int fb = pp / p.P;
                    const int fp = pp - fb * p.P;
                    row = (k < p.k1) ? p.x1 + (size_t)fb * p.x1_bs + (size_t)k * p.P
                                     : p.x2 + (size_t)fb * p.x2_bs + (size_t)(k - p.k1) * p.P;
                    if (pp + 3 < ncols) v = *reinterpret_cast<const float4*>(row + fp);
                } else {
                    row = (k < p.k1) ? x1 + (size_t)k * p.P : x2 + (size_t)(k - p.k1) * p.P;
                    if (vecP && pp + 3 < p.P) {
                        v = *reinterpret_cast<const float4*>(row + pp);
                    } else {
                        if (pp < p.P) v.x = row[pp];
                        if (pp + 1 < p.P) v.y = row[pp + 1];
                        if (pp + 2 < p.P) v.z = row[pp + 2];
                        if (pp + 3 < p.P) v.w = row[pp + 3];
                    }
                }
            }
            rb[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        constexpr int A_TOTAL = BK * BM / 4;
#pragma unroll
        for (int i = 0; i < (A_F4 > 0 ? A_F4 : 1); ++i) {
            const int f = tid + i * BLK;
            if (f < A_TOTAL) {
                const int kr = f / (BM / 4), mc = (f % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(&As[buf][kr][mc]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * BLK;
            const int kr = f / (BN / 4), nc = (f % (BN / 4)) * 4;
            *reinterpret_cast<float4*>(&Bs[buf][kr][nc]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kh = lane >> 5;       // which of the 2 k of an MFMA this lane feeds
    const int l31 = lane & 31;
    const int am = wm * (TM * 32) + l31;
    const int bn = wn * (TN * 32) + l31;

    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    if (kbeg + BK < kend) load_tiles(kbeg + BK);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
        // stage k+1 goes to the other LDS buffer (its last readers passed the previous barrier),
        // the global loads of stage k+2 stay in flight under this stage's MFMAs
        if (k0 + BK < kend) store_tiles(buf ^ 1);
        if (k0 + 2 * BK < kend) load_tiles(k0 + 2 * BK);
        // fragment double buffering: the LDS reads of step kk+2 are in flight under the MFMAs of kk
        float a[2][TM], bb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][kh][am + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bb[0][j] = Bs[buf][kh][bn + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = As[buf][kk + 2 + kh][am + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[nxt][j] = Bs[buf][kk + 2 + kh][bn + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ahead of this step's MFMAs
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // epilogue: bias + gathered term + activation; for a fixed accumulator register the 32 lanes
    // of a half-wave hold 32 consecutive p of one output row -> 128-byte coalesced stores
    if constexpr (ATT) {
        // out = pooled [B, cout, P/16]; every 32-column MFMA tile holds two complete points.
        // softmax_k(a) . f = sum_k f_k 2^((a_k - max) log2 e) / sum_k 2^((a_k - max) log2 e): one v_exp_f32 and
        // one v_rcp_f32 per element (both ~1 ulp) instead of the libm expf and an IEEE division; 32-bit
        // element offsets (the entry point bounds d * P).
        float* out = p.out + (size_t)b * p.out_bs;
        const int npts = p.P >> 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = p0 + wn * (TN * 32) + j * 32 + l31;
            const bool in = col < p.P;                       // P % 16 == 0: a row of 16 lanes is in or out together
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const bool ok = in && m < p.cout;
                    const float a = ok ? acc[i][j][r] : 0.f;
                    float f = 0.f;
                    if (ok) {
                        const bool first = m < p.k1;
                        const float* src = first ? x1 : x2;
                        f = src[(unsigned)((first ? m : m - p.k1) * p.P + col)];
                    }
                    const float e = __builtin_amdgcn_exp2f((a - row16_max(a)) * 1.44269504088896341f);
                    const float pooled = row16_sum(f * e) * __builtin_amdgcn_rcpf(row16_sum(e));
                    if (ok && (l31 & 15) == 0) out[(unsigned)(m * npts + (col >> 4))] = pooled;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = p0 + wn * (TN * 32) + j * 32 + l31;
        if (col >= ncols) continue;
        if (FLAT && p.part) {       // K is split: raw partial sums, reduced by shared_mlp_reduce_kernel
            float* dst = p.part + (size_t)bz * p.cout * ncols + col;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m < p.cout) dst[(size_t)m * ncols] = acc[i][j][r];
                }
            continue;
        }
        const int fb = FLAT ? col / p.P : b;
        const int pp = FLAT ? col - fb * p.P : col;
        float* out = p.out + (size_t)fb * p.out_bs;
        const float* yg = p.yg ? p.yg + (size_t)fb * p.yg_bs : nullptr;
        long long gi = 0;
        if (yg) {
            gi = p.idx64 ? static_cast<const long long*>(p.gidx)[(size_t)fb * p.P + pp]
                         : (long long)static_cast<const int*>(p.gidx)[(size_t)fb * p.P + pp];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.cout) {
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[m];
                    if (yg) v += yg[(size_t)m * p.py + gi];
                    out[(size_t)m * p.P + pp] = activate(v, p.act);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pipelined variant for the MFMA-bound launches (per-frame tiles, 16-byte aligned rows).
//
// PMC of shared_mlp_kernel<128> on the 1024->1024 layer showed the MFMA pipe only ~60 % busy with
// three workgroups per CU: the tile refill (bounds-checked loads + LDS stores, full of exec-mask
// branches) is a separate phase of the loop that the compiler cannot mix into the MFMA stream, and
// co-resident workgroups run in lock-step, so their refill phases coincide.
// (measured alternatives that did not help: BK = 32 with two workgroups per CU, s_setprio around the
// MFMA groups, and a third-generation k-loop with LDS-direct loads into three LDS stages + hand-placed vmcnt
// (validated on hardware at the start of round 2, profiles/r02_shared_mlp_lds_ab.txt: bit-identical, within +-3 % on
// the large layers) -- removed again.  The point-major path (csrc/mlp_pm.hip) superseded this kernel as the default.)  Here the k-loop body
// is ONE basic block: operands come through buffer loads (hardware range check returns 0 outside
// [0, num_records) -> no branches for the K tail or the end of a source), the refill is issued
// unconditionally (the surplus loads of the last steps are out of range = free zeros) and is
// spread over the k-steps of the MFMA stream.
// Out-of-tile columns/rows (m >= cout, p >= P) read neighbouring in-range data instead of zeros;
// they only feed accumulator elements that are never stored.
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buffer_load_f4(__amdgpu_buffer_rsrc_t rs, int byte_offset)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_offset, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int BM>
__global__ void __launch_bounds__(BLK)
shared_mlp_pipe_kernel(const MlpParams p)
{
    constexpr int WM = BM == 128 ? 2 : 1;
    constexpr int WN = 4 / WM;
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int A_F4 = BK * BM / 4 / BLK;          // 128: 2, 64: 1
    constexpr int B_F4 = BK * BN / 4 / BLK;          // 2
    static_assert(A_F4 >= 1, "BM >= 64");
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ncol_tiles = p.col_tiles;
    if ((int)blockIdx.x >= ncol_tiles * p.nz) return;
    const int b = blockIdx.x / ncol_tiles;
    const int m0 = blockIdx.y * BM;
    const int p0 = (blockIdx.x - b * ncol_tiles) * BN;
    const int K = p.k1 + p.k2;

    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, K * p.cout * 4, 0x00020000);
    const float* x1 = p.x1 + (size_t)b * p.x1_bs;
    const float* x2 = p.x2 ? p.x2 + (size_t)b * p.x2_bs : p.x1;
    const int rec1 = p.k1 * p.P * 4, rec2 = p.x2 ? p.k2 * p.P * 4 : 0;

    int a_vo[A_F4], b_vo[B_F4], a_lds[A_F4], b_lds[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int f = tid + i * BLK;
        const int kr = f / (BM / 4), mc = (f % (BM / 4)) * 4;
        a_vo[i] = (kr * p.cout + m0 + mc) * 4;
        a_lds[i] = kr * BM + mc;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int f = tid + i * BLK;
        const int kr = f / (BN / 4), nc = (f % (BN / 4)) * 4;
        b_vo[i] = (kr * p.P + p0 + nc) * 4;
        b_lds[i] = kr * BN + nc;
    }
    float4 ra[A_F4], rb[B_F4];
    auto load_tiles = [&](int k0) {                  // k0 is a multiple of BK; k1 % BK == 0 when there is an x2
        const bool second = k0 >= p.k1;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(second ? x2 : x1), 0, second ? rec2 : rec1, 0x00020000);
        const int a_so = k0 * p.cout * 4;
        const int b_so = (second ? k0 - p.k1 : k0) * p.P * 4;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) ra[i] = buffer_load_f4(rs_w, a_vo[i] + a_so);
#pragma unroll
        for (int i = 0; i < B_F4; ++i) rb[i] = buffer_load_f4(rs_x, b_vo[i] + b_so);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(&As[buf][0][0] + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) *reinterpret_cast<float4*>(&Bs[buf][0][0] + b_lds[i]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kh = lane >> 5;
    const int l31 = lane & 31;
    const int am = wm * (TM * 32) + l31;
    const int bn = wn * (TN * 32) + l31;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    load_tiles(BK);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
        float a[2][TM], bb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][kh][am + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bb[0][j] = Bs[buf][kh][bn + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = As[buf][kk + 2 + kh][am + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[nxt][j] = Bs[buf][kk + 2 + kh][bn + j * 32];
            }
            if (kk == 2) store_tiles(buf ^ 1);       // stage k+1 (loaded one iteration ago) -> other LDS buffer
            if (kk == 4) load_tiles(k0 + 2 * BK);     // stage k+2 -> the staging registers just freed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    const float* yg = p.yg ? p.yg + (size_t)b * p.yg_bs : nullptr;
    float* out = p.out + (size_t)b * p.out_bs;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = p0 + wn * (TN * 32) + j * 32 + l31;
        if (col >= p.P) continue;
        long long gi = 0;
        if (yg) {
            gi = p.idx64 ? static_cast<const long long*>(p.gidx)[(size_t)b * p.P + col]
                         : (long long)static_cast<const int*>(p.gidx)[(size_t)b * p.P + col];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.cout) {
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[m];
                    if (yg) v += yg[(size_t)m * p.py + gi];
                    out[(size_t)m * p.P + col] = activate(v, p.act);
                }
            }
        }
    }
}

// sums the K-split partial slabs and applies the epilogue; one lane per output element
__global__ void __launch_bounds__(BLK)
shared_mlp_reduce_kernel(const MlpParams p, int nsplit)
{
    const size_t ncols = (size_t)p.nb * p.P;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= ncols * p.cout) return;
    const int m = (int)(t / ncols);
    const size_t col = t - (size_t)m * ncols;
    float v = 0.f;
    for (int z = 0; z < nsplit; ++z) v += p.part[((size_t)z * p.cout + m) * ncols + col];
    const int fb = (int)(col / p.P);
    const int pp = (int)(col - (size_t)fb * p.P);
    if (p.bias) v += p.bias[m];
    if (p.yg) {
        const long long gi = p.idx64 ? static_cast<const long long*>(p.gidx)[(size_t)fb * p.P + pp]
                                     : (long long)static_cast<const int*>(p.gidx)[(size_t)fb * p.P + pp];
        v += p.yg[(size_t)fb * p.yg_bs + (size_t)m * p.py + gi];
    }
    p.out[(size_t)fb * p.out_bs + (size_t)m * p.P + pp] = activate(v, p.act);
}

struct MlpPlan {
    bool flat;
    int nsplit, kchunk;
};

// FFB6D_MLP_PIPE=0 in the environment routes everything through the first-generation kernel
// (A/B measurements, scripts/bench_mlp.py)
int mlp_pipe_enabled()
{
    static const int v = [] {
        const char* e = getenv("FFB6D_MLP_PIPE");
        return e ? atoi(e) : 1;
    }();
    return v;
}

MlpPlan plan_mlp(int64_t B, int64_t cout, int64_t K, int64_t P)
{
    MlpPlan pl{false, 1, 0};
    // small per-frame P: per-frame tiles would be mostly padding and too few to fill 256 CUs
    if (P >= 2048 || (P & 3) != 0) return pl;
    pl.flat = true;
    const int64_t bm = cout > 64 ? 128 : (cout > 32 ? 64 : 32);
    const int64_t blocks = ceil_div(B * P, BN) * ceil_div(cout, bm);
    int64_t ns = ceil_div(512, blocks);                  // aim at ~2 blocks per CU
    const int64_t max_by_k = K / 64 > 0 ? K / 64 : 1;    // keep >= 4 k-steps per split
    if (ns > max_by_k) ns = max_by_k;
    if (ns > 32) ns = 32;
    if (ns < 1) ns = 1;
    int64_t kc = ceil_div(ceil_div(K, ns), BK) * BK;
    pl.nsplit = (int)ceil_div(K, kc);
    pl.kchunk = (int)kc;
    return pl;
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

extern "C" int ffb6d_att_score_pool_f32(const float* wt, const float* x1, int64_t k1, const float* x2, int64_t k2,
                                        float* out, int64_t B, int64_t N, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(K == 16, "att_score_pool: K must be 16 (got %d)", K);
    FFB6D_REQUIRE(B >= 0 && N >= 0 && k1 >= 1 && k2 >= 0, "att_score_pool: bad shape");
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(wt && x1 && out && ((k2 == 0) == (x2 == nullptr)), "att_score_pool: null pointer");
    const int64_t d = k1 + k2, P = N * 16;
    FFB6D_REQUIRE(P < (1LL << 31) && d < (1 << 20) && B < 65536 && d * P < (1LL << 31), "att_score_pool: too large");
    MlpParams p;
    p.wt = wt; p.bias = nullptr; p.x1 = x1; p.x2 = x2; p.yg = nullptr; p.gidx = nullptr; p.out = out;
    p.x1_bs = k1 * P; p.x2_bs = k2 * P; p.yg_bs = 0; p.out_bs = d * N;
    p.k1 = (int)k1; p.k2 = (int)k2; p.cout = (int)d; p.P = (int)P; p.py = 0; p.act = 0; p.idx64 = 0;
    p.nb = (int)B; p.kchunk = (int)d; p.part = nullptr;
    hipStream_t st = as_stream(stream);
    const unsigned gx = (unsigned)ceil_div(P, BN);
    p.col_tiles = (int)gx; p.nz = (int)B;
    const unsigned gxz = (gx * (unsigned)B + 7u) / 8u * 8u;
    if (d > 64) {
        hipLaunchKernelGGL((shared_mlp_kernel<128, false, true>), dim3(gxz, (unsigned)ceil_div(d, 128), 1), dim3(BLK), 0, st, p);
    } else if (d > 32) {
        hipLaunchKernelGGL((shared_mlp_kernel<64, false, true>), dim3(gxz, 1, 1), dim3(BLK), 0, st, p);
    } else {
        hipLaunchKernelGGL((shared_mlp_kernel<32, false, true>), dim3(gxz, 1, 1), dim3(BLK), 0, st, p);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" size_t ffb6d_shared_mlp_workspace_bytes(int64_t B, int64_t cout, int64_t K, int64_t P)
{
    if (B <= 0 || cout <= 0 || K <= 0 || P <= 0) return 0;
    const MlpPlan pl = plan_mlp(B, cout, K, P);
    return pl.nsplit > 1 ? (size_t)pl.nsplit * cout * B * P * sizeof(float) : 0;
}

extern "C" int ffb6d_shared_mlp_f32(const float* wt, const float* bias, const float* x1, int64_t k1,
                                    int64_t x1_batch_stride, const float* x2, int64_t k2,
                                    int64_t x2_batch_stride, const float* ygather, const void* gidx,
                                    int idx_bits, int64_t py, int64_t yg_batch_stride, float* out,
                                    int64_t out_batch_stride, int64_t B, int64_t cout, int64_t P, int act,
                                    void* workspace, size_t workspace_bytes, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && cout >= 1 && P >= 0 && k1 >= 1 && k2 >= 0, "shared_mlp: bad shape");
    FFB6D_REQUIRE(act >= 0 && act <= 2, "shared_mlp: act must be 0 (none), 1 (relu) or 2 (leaky 0.2)");
    FFB6D_REQUIRE(P < (1LL << 31) && cout < (1 << 20) && k1 + k2 < (1 << 20) && B < 65536, "shared_mlp: too large");
    if (B == 0 || P == 0) return FFB6D_OK;
    FFB6D_REQUIRE(wt && x1 && out, "shared_mlp: null pointer");
    FFB6D_REQUIRE((k2 == 0) == (x2 == nullptr), "shared_mlp: x2 and k2 must come together");
    FFB6D_REQUIRE((ygather == nullptr) == (gidx == nullptr), "shared_mlp: gather term needs values and indices");
    FFB6D_REQUIRE(!ygather || idx_bits == 32 || idx_bits == 64, "shared_mlp: idx_bits must be 32 or 64");
    MlpParams p;
    p.wt = wt; p.bias = bias; p.x1 = x1; p.x2 = x2; p.yg = ygather; p.gidx = gidx; p.out = out;
    p.x1_bs = x1_batch_stride; p.x2_bs = x2_batch_stride; p.yg_bs = yg_batch_stride; p.out_bs = out_batch_stride;
    p.k1 = (int)k1; p.k2 = (int)k2; p.cout = (int)cout; p.P = (int)P; p.py = (int)py; p.act = act;
    p.idx64 = idx_bits == 64;
    hipStream_t st = as_stream(stream);
    MlpPlan pl = plan_mlp(B, cout, k1 + k2, P);
    if (pl.flat && ((x1_batch_stride | x2_batch_stride) & 3 ||
                    ((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2)) & 15)))
        pl = MlpPlan{false, 1, 0};            // flat mode needs 16-byte aligned rows
    p.nb = (int)B; p.kchunk = pl.flat ? pl.kchunk : (int)(k1 + k2); p.part = nullptr;
    if (pl.flat && pl.nsplit > 1) {
        const size_t need = (size_t)pl.nsplit * cout * B * P * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return set_error(FFB6D_ERR_WORKSPACE, "shared_mlp: workspace of %zu bytes required, got %zu", need,
                             workspace ? workspace_bytes : (size_t)0);
        p.part = static_cast<float*>(workspace);
    }
    const unsigned gx = (unsigned)ceil_div(pl.flat ? B * P : P, BN);
    const unsigned gz = pl.flat ? (unsigned)pl.nsplit : (unsigned)B;
    p.col_tiles = (int)gx; p.nz = (int)gz;
    const unsigned gxz = (gx * gz + 7u) / 8u * 8u;       // multiple of the XCD count
    // pipelined kernel: per-frame tiles, 16-byte aligned rows, sources switching on a BK boundary,
    // byte offsets that fit the 32-bit buffer addressing
    const int64_t K = k1 + k2;
    const bool pipe = !pl.flat && cout > 32 && mlp_pipe_enabled() && (P & 3) == 0 && (cout & 3) == 0 &&
                      ((x1_batch_stride | x2_batch_stride) & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2) |
                        reinterpret_cast<uintptr_t>(wt)) & 15) == 0 &&
                      (k2 == 0 || k1 % BK == 0) && (K + 2 * BK) * P * 4 < (1LL << 31) &&
                      (K + 2 * BK) * cout * 4 < (1LL << 31);
    if (pipe) {
        if (cout > 64)
            hipLaunchKernelGGL((shared_mlp_pipe_kernel<128>), dim3(gxz, (unsigned)ceil_div(cout, 128), 1), dim3(BLK), 0, st, p);
        else
            hipLaunchKernelGGL((shared_mlp_pipe_kernel<64>), dim3(gxz, 1, 1), dim3(BLK), 0, st, p);
        FFB6D_LAUNCH_CHECK();
        return FFB6D_OK;
    }
#define FFB6D_LAUNCH_MLP(BMV)                                                                              \
    do {                                                                                                   \
        const dim3 grid(gxz, (unsigned)ceil_div(cout, BMV), 1);                                            \
        if (pl.flat) hipLaunchKernelGGL((shared_mlp_kernel<BMV, true>), grid, dim3(BLK), 0, st, p);        \
        else hipLaunchKernelGGL((shared_mlp_kernel<BMV, false>), grid, dim3(BLK), 0, st, p);               \
    } while (0)
    if (cout > 64) FFB6D_LAUNCH_MLP(128);
    else if (cout > 32) FFB6D_LAUNCH_MLP(64);
    else FFB6D_LAUNCH_MLP(32);
#undef FFB6D_LAUNCH_MLP
    FFB6D_LAUNCH_CHECK();
    if (p.part) {
        const size_t total = (size_t)B * P * cout;
        hipLaunchKernelGGL(shared_mlp_reduce_kernel, dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, st, p,
                           pl.nsplit);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
