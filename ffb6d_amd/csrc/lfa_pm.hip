// ffb6d_amd/csrc/lfa_pm.hip -- one half of RandLA-Net's local feature aggregation as ONE kernel, for gfx950.
//
// Reference: Building_block.forward, ffb6d/models/RandLA/RandLANet.py:196-214, with relative_pos_encoding :216-223,
// gather_neighbour :225-234 and Att_pooling.forward :243-250.  Per point n with its 16 neighbours k:
//
//   half 1 (MODE 1)                                              half 2 (MODE 2)
//     e     = [|p-q|, p-q, p, q]            (10 channels)          e, g1 as in half 1 (recomputed, never stored)
//     g1    = act(W1 e + b1)                lfa.mlp1               g2  = act(W2 g1 + b2)                   lfa.mlp2
//     S     = [ f[nei[n,k]] | g1 ]          feature set, d wide    S   = [ f_agg[nei[n,k]] | g2 ]
//     A     = S Wfc^T                       att_pooling_1.fc       A   = S Wfc^T                           att_pooling_2.fc
//     pool  = sum_k S * softmax_k(A)                               pool likewise
//     out   = act(Wm pool + bm)   [d/2]     att_pooling_1.mlp      out = act(Wm pool + bm)   [d]           att_pooling_2.mlp
//
// The unfused path (csrc/posenc.hip -> att_pool_pm -> mlp_pm) writes the per-pair tensor g [B,N,16,d/2] to HBM and reads it
// back two to four times: ~500 MB per step at the first level for data that is a pure function of 14 MB of coordinates and
// indices.  Here a workgroup owns P consecutive points and builds their 16 P pair rows ONCE, in LDS:
//
//   * neighbour tiles staged in LDS: the gathered feature rows arrive as whole rows (consecutive lanes fetch consecutive 16-byte
//     chunks of one row), the position-encoding half of a pair row is computed on the vector ALU (10-term FMA chain, the
//     arithmetic of csrc/posenc_body.h) and -- half 2 -- pushed through lfa.mlp2 on the matrix cores; both land in one image
//     [16 P rows][d] whose row order inside a 32-row tile is the slot order of att_pool_pm_kernel
//         slot rho -> point (rho >> 2) & 1 of the tile, neighbour (rho & 3) + 4 * (rho >> 3),
//     so that accumulator register r of lane l of the score GEMM is neighbour r of point (l >> 5), channel (l & 31): the softmax
//     over the neighbourhood and the weighted sum are in-lane arithmetic, the feature values they multiply come out of the image;
//   * the pooled rows meet in a second small image and the output MLP runs on them before anything is written: the kernel reads
//     coordinates, indices and point rows and writes point rows -- nothing per pair touches HBM;
//   * weights stream from L2 as MFMA fragments (buffer loads, three register stages); the four waves split the (row tile, channel
//     tile) grid of each GEMM so that every wave carries four accumulator tiles;
//   * workgroup -> points: XCD x (blockIdx % 8) walks the x-th eighth of the point groups, i.e. one frame of a batch of 8 per
//     XCD: the rows a gather can address (one frame) stay in that XCD's L2.
//
// Algorithmic bytes at the kernel's boundary: 12 B N (xyz) + idx + esz B N d/2 (point rows in) + esz B N cout (out) + weights.
// Algorithmic flops: 2*16 B N d^2 (scores) + 2*16 B N 10 d/2 (mlp1) [+ 2*16 B N (d/2)^2 (mlp2)] + 2 B N d cout (output MLP).
#include <algorithm>

#include "common.h"
#include "ffb6d_ops.h"
#include "mfma_pm.h"

namespace ffb6d {
namespace {

using namespace pm;

struct LfaParams {
    const float* xyz;     // [B * N, 3]
    const void* nei;      // [B * N * 16] int32 / int64 neighbour indices inside the frame
    const void* f;        // [B * N, ldf] point rows of T, first d/2 elements used
    const float* w1;      // [d/2, ldw1] fp32 (BatchNorm folded), columns 0..9 used
    const float* b1;      // [d/2] fp32
    const void* w2;       // MODE 2: [d/2, d/2] of T
    const float* b2;      // MODE 2: [d/2] fp32
    const void* wfc;      // [d, d] of T (no bias)
    const void* wm;       // [cout, d] of T
    const float* bm;      // [cout] fp32
    void* out;            // [B * N, ldo] of T
    int npts, N, ldf, ldo, ldw1, idx64;
    float slope1, slope2, slopem;      // act(v) = max(v, slope * v)
    int n_grp;            // point groups (workgroups with work)
};

template <typename T> __device__ __forceinline__ u32x4 pack_chunk(const float (&v)[16 / El<T>::SZ])
{
    if constexpr (El<T>::SZ == 4) {
        return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    } else {
        bf16x8 b;
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[e];        // round to nearest even
        return __builtin_bit_cast(u32x4, b);
    }
}

// acc[TA][TB] += A B over NSTEPS 32-byte steps; la(step, i) / lb(step, j) deliver the 16-byte fragment of this lane.  Three
// register stages: the fragments of step s + 2 are requested before step s is multiplied (global fragments need the depth,
// LDS fragments do not mind it); sched_barrier pins the order (hipcc otherwise sinks the loads below the MFMA groups).
template <typename T, int NSTEPS, int TA, int TB, typename LA, typename LB>
__device__ __forceinline__ void gemm_steps(f32x16 (&acc)[TA][TB], LA&& la, LB&& lb)
{
    u32x4 a0[TA], a1[TA], a2[TA], b0[TB], b1[TB], b2[TB];
    auto load = [&](int s, u32x4 (&a)[TA], u32x4 (&b)[TB]) {
        const int sc = s < NSTEPS ? s : NSTEPS - 1;             // surplus prefetch: re-read the last step (never used)
#pragma unroll
        for (int i = 0; i < TA; ++i) a[i] = la(sc, i);
#pragma unroll
        for (int j = 0; j < TB; ++j) b[j] = lb(sc, j);
    };
#define FFB6D_PIN() __builtin_amdgcn_sched_barrier(0)
    load(0, a0, b0);
    load(1, a1, b1);
    int st = 0;
    for (; st + 3 <= NSTEPS; st += 3) {
        load(st + 2, a2, b2);                  FFB6D_PIN();
        mfma_step<T, TA, TB>(acc, a0, b0);     FFB6D_PIN();
        load(st + 3, a0, b0);                  FFB6D_PIN();
        mfma_step<T, TA, TB>(acc, a1, b1);     FFB6D_PIN();
        load(st + 4, a1, b1);                  FFB6D_PIN();
        mfma_step<T, TA, TB>(acc, a2, b2);     FFB6D_PIN();
    }
#undef FFB6D_PIN
    if (st < NSTEPS) mfma_step<T, TA, TB>(acc, a0, b0);
    if (st + 1 < NSTEPS) mfma_step<T, TA, TB>(acc, a1, b1);
}

template <int TA, int TB> __device__ __forceinline__ void zero(f32x16 (&acc)[TA][TB])
{
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

template <typename T, int D, int P> struct LfaGeom {
    static constexpr int SZ = El<T>::SZ;
    static constexpr int RS = D * SZ + 16;            // image row stride: an odd multiple of 16 bytes (conflict-free ds_read_b128)
    static constexpr int ROWS = 16 * P;               // pair rows of a workgroup
    static constexpr int S_BYTES = ROWS * RS;         // pair image
    static constexpr int PL_BYTES = P * RS;           // pooled rows
    static constexpr int LDS = S_BYTES + PL_BYTES + 128;
};

template <typename T, int D, int MODE, int P>
__global__ void __launch_bounds__(BLK, 2)
lfa_pm_kernel(const LfaParams p)
{
    using G = LfaGeom<T, D, P>;
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;                    // k per 32-byte step
    constexpr int H = D / 2;
    constexpr int VL = 16 / SZ;                       // channels per 16-byte chunk
    constexpr int CPR = H / VL;                       // chunks of half a pair row
    constexpr int RS = G::RS, ROWS = G::ROWS;
    constexpr int NRT = P / 2;                        // row tiles (2 points x 16 neighbours)
    constexpr int NCT = D / 32;                       // channel tiles of the score GEMM
    constexpr int NCH = ROWS * CPR / BLK;             // chunks per thread of the gather / of the encoding MLP
    constexpr int RPI = BLK / CPR;                    // pair rows covered by one chunk per thread
    constexpr int COUT = MODE == 1 ? H : D;
    constexpr int OOB = 0x7ffffff0;
    static_assert(D % 32 == 0 && P % 2 == 0 && P <= 32, "tile geometry");
    static_assert(H % KSTEP == 0, "half a pair row must be whole 32-byte steps");
    static_assert(BLK % CPR == 0 && (ROWS * CPR) % BLK == 0 && NCH >= 1, "chunks must divide evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [pair image | pooled rows]
    unsigned char* const S = lds;
    unsigned char* const PL = lds + G::S_BYTES;

    // XCD x walks the x-th eighth of the point groups
    const int per_xcd = (p.n_grp + 7) >> 3;
    const int g = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || g >= p.n_grp) return;
    const int n0 = g * P;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---------------------------------------------------------------------------------------------------------------
    // 1. pair rows: gathered point rows (registers for now) + position encoding -> mlp1 on the vector ALU
    // ---------------------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(p.f, (unsigned)p.npts * (unsigned)p.ldf * SZ);
    const int col = tid % CPR, row0 = tid / CPR;
    u32x4 fch[NCH];
    int lrow[NCH];                                    // image row (slot order) of chunk i
    {
        float w1[VL][10], b1[VL];
#pragma unroll
        for (int e = 0; e < VL; ++e) {
            const float* wr = p.w1 + (size_t)(col * VL + e) * p.ldw1;
#pragma unroll
            for (int t = 0; t < 10; ++t) w1[e][t] = wr[t];
            b1[e] = p.b1[col * VL + e];
        }
        int src[NCH];                                 // source point row (all frames) of chunk i, -1 = past the last point
        int self[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int row = row0 + i * RPI, pp = row >> 4, nb = row & 15;
            const int n = n0 + pp;
            lrow[i] = (pp >> 1) * 32 + ((nb & 3) | ((pp & 1) << 2) | ((nb >> 2) << 3));
            src[i] = -1;
            self[i] = 0;
            if (n < p.npts) {
                const size_t pair = (size_t)n * 16 + nb;
                const int nbi = p.idx64 ? (int)static_cast<const long long*>(p.nei)[pair] : static_cast<const int*>(p.nei)[pair];
                src[i] = (n / p.N) * p.N + nbi;
                self[i] = n;
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            fch[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, src[i] >= 0 ? src[i] * p.ldf * SZ + col * 16 : OOB, 0, 0);
        // encoding + mlp1 (the arithmetic of csrc/posenc_body.h: separately rounded products and sums, IEEE sqrt, FMA chain in the
        // order bias, dis, dx, dy, dz, p, q); points past the end compute on point 0 -- finite, never stored
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float* pv = p.xyz + (size_t)self[i] * 3;
            const float* qv = p.xyz + (size_t)(src[i] >= 0 ? src[i] : 0) * 3;
            const float px = pv[0], py = pv[1], pz = pv[2];
            const float qx = qv[0], qy = qv[1], qz = qv[2];
            const float dx = px - qx, dy = py - qy, dz = pz - qz;
            const float s2 = ((dx * dx) + (dy * dy)) + (dz * dz);       // -ffp-contract=off: no fusion
            const float enc[10] = {sqrtf(s2), dx, dy, dz, px, py, pz, qx, qy, qz};
            float o[VL];
#pragma unroll
            for (int e = 0; e < VL; ++e) {
                float v = b1[e];
#pragma unroll
                for (int t = 0; t < 10; ++t) v = fmaf(w1[e][t], enc[t], v);
                o[e] = activate(v, p.slope1);
            }
            // half 1: this IS the second half of the pair row; half 2: parked in the first half until mlp2 has consumed it
            *reinterpret_cast<u32x4*>(S + lrow[i] * RS + (MODE == 1 ? H * SZ : 0) + col * 16) = pack_chunk<T>(o);
        }
    }

    if constexpr (MODE == 2) {
        // -----------------------------------------------------------------------------------------------------------
        // 2. lfa.mlp2 on the matrix cores: channels = MFMA rows (weights from L2), pair rows = MFMA columns (image, first half)
        //    -> second half of the image.  A lane ends up with 4 x 4 consecutive channels of ONE pair row.
        // -----------------------------------------------------------------------------------------------------------
        __syncthreads();
        constexpr int NOT2 = H / 32 > 0 ? H / 32 : 1;                  // output channel tiles (H = 16: half a tile)
        constexpr int WR2 = 4 / NOT2;                                  // waves along the row tiles
        constexpr int RT2 = (NRT + WR2 - 1) / WR2;                     // row tiles per wave
        const int ot = wave % NOT2, wr = wave / NOT2;
        const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(p.w2, (unsigned)(H * H * SZ));
        f32x16 acc[1][RT2];
        zero(acc);
        const int w_vo = (ot * 32 + l31) * H * SZ + 16 * kh;           // rows past H: out of range -> zeros
        gemm_steps<T, H / KSTEP, 1, RT2>(
            acc,
            [&](int s, int) { return __builtin_amdgcn_raw_buffer_load_b128(rs_w2, w_vo + s * 32, 0, 0); },
            [&](int s, int j) {
                const int rt = min(wr + WR2 * j, NRT - 1);
                return *reinterpret_cast<const u32x4*>(S + (rt * 32 + l31) * RS + s * 32 + 16 * kh);
            });
#pragma unroll
        for (int j = 0; j < RT2; ++j) {
            const int rt = wr + WR2 * j;
            if (rt >= NRT) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = ot * 32 + 8 * q + 4 * kh;
                if (ch >= H) continue;
                const float4 b4 = *reinterpret_cast<const float4*>(p.b2 + ch);
                const float4 v = make_float4(activate(acc[0][j][4 * q] + b4.x, p.slope2), activate(acc[0][j][4 * q + 1] + b4.y, p.slope2),
                                             activate(acc[0][j][4 * q + 2] + b4.z, p.slope2), activate(acc[0][j][4 * q + 3] + b4.w, p.slope2));
                El<T>::st4(reinterpret_cast<T*>(S + (rt * 32 + l31) * RS) + H + ch, v);
            }
        }
        __syncthreads();                              // every wave is done reading the parked mlp1 rows
    }
    // gathered point rows -> first half of the image
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(S + lrow[i] * RS + col * 16) = fch[i];
    __syncthreads();

    // ---------------------------------------------------------------------------------------------------------------
    // 3. scores A = S Wfc^T (pair rows = MFMA rows out of the image, channels = MFMA columns, weights from L2), then in-lane
    //    softmax over the 16 neighbours and the weighted sum -> pooled rows
    // ---------------------------------------------------------------------------------------------------------------
    {
        constexpr int WC = NCT < 4 ? NCT : 4;                          // waves along the channel tiles
        constexpr int WR = 4 / WC;
        constexpr int TN = NCT / WC;
        constexpr int TM = (NRT + WR - 1) / WR;
        const int wc = wave % WC, wr = wave / WC;
        const __amdgpu_buffer_rsrc_t rs_fc = make_rsrc(p.wfc, (unsigned)(D * D * SZ));
        f32x16 acc[TM][TN];
        zero(acc);
        const int w_vo = (wc * TN * 32 + l31) * D * SZ + 16 * kh;
        gemm_steps<T, D / KSTEP, TM, TN>(
            acc,
            [&](int s, int i) {
                const int rt = min(wr + WR * i, NRT - 1);
                return *reinterpret_cast<const u32x4*>(S + (rt * 32 + l31) * RS + s * 32 + 16 * kh);
            },
            [&](int s, int j) { return __builtin_amdgcn_raw_buffer_load_b128(rs_fc, w_vo + j * 32 * D * SZ + s * 32, 0, 0); });
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rt = wr + WR * i;
            if (rt >= NRT) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ch = (wc * TN + j) * 32 + l31;
                const unsigned char* sp = S + (rt * 32 + (kh << 2)) * RS + ch * SZ;      // slot of (point kh, neighbour 0)
                float fv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) fv[r] = El<T>::ld(reinterpret_cast<const T*>(sp + ((r & 3) | ((r >> 2) << 3)) * RS));
                float m = acc[i][j][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                float num = 0.f, den = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f((acc[i][j][r] - m) * 1.44269504088896341f);
                    den += e;
                    num = fmaf(fv[r], e, num);
                }
                El<T>::st(reinterpret_cast<T*>(PL + (2 * rt + kh) * RS) + ch, num * __builtin_amdgcn_rcpf(den));
            }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------------------------------------------------------
    // 4. output MLP on the pooled rows: channels = MFMA rows (weights from L2), points = MFMA columns (pooled image)
    // ---------------------------------------------------------------------------------------------------------------
    {
        constexpr int NOT = COUT / 32 > 0 ? COUT / 32 : 1;
        constexpr int TMO = (NOT + 3) / 4;
        if (wave >= NOT) return;
        const __amdgpu_buffer_rsrc_t rs_wm = make_rsrc(p.wm, (unsigned)(COUT * D * SZ));
        f32x16 acc[TMO][1];
        zero(acc);
        const int w_vo = (wave * 32 + l31) * D * SZ + 16 * kh;         // tiles wave, wave + 4; rows past COUT: out of range -> zeros
        const unsigned char* xp = PL + min(l31, P - 1) * RS + 16 * kh;
        gemm_steps<T, D / KSTEP, TMO, 1>(
            acc,
            [&](int s, int i) {
                return __builtin_amdgcn_raw_buffer_load_b128(rs_wm, wave + 4 * i < NOT ? w_vo + i * 128 * D * SZ + s * 32 : OOB, 0, 0);
            },
            [&](int s, int) { return *reinterpret_cast<const u32x4*>(xp + s * 32); });
        const int n = n0 + l31;
        if (l31 >= P || n >= p.npts) return;
        T* orow = static_cast<T*>(p.out) + (size_t)n * p.ldo;
#pragma unroll
        for (int i = 0; i < TMO; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = (wave + 4 * i) * 32 + 8 * q + 4 * kh;
                if (ch >= COUT) continue;
                const float4 b4 = *reinterpret_cast<const float4*>(p.bm + ch);
                El<T>::st4(orow + ch, make_float4(activate(acc[i][0][4 * q] + b4.x, p.slopem), activate(acc[i][0][4 * q + 1] + b4.y, p.slopem),
                                                  activate(acc[i][0][4 * q + 2] + b4.z, p.slopem), activate(acc[i][0][4 * q + 3] + b4.w, p.slopem)));
            }
        }
    }
}

template <typename T, int D, int MODE, int P>
void launch_lfa(LfaParams& p, hipStream_t st)
{
    using G = LfaGeom<T, D, P>;
    p.n_grp = (int)ceil_div(p.npts, P);
    const void* fn = reinterpret_cast<const void*>(&lfa_pm_kernel<T, D, MODE, P>);
    static const hipError_t attr = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    (void)attr;
    const unsigned grid = (unsigned)(ceil_div(p.n_grp, 8) * 8);
    hipLaunchKernelGGL((lfa_pm_kernel<T, D, MODE, P>), dim3(grid), dim3(BLK), G::LDS, st, p);
}

// points per workgroup: 16 P d elements of pair image = 64 KB (fp32) whatever the level; `small` halves it (more, smaller
// workgroups: the deep levels have few points)
template <typename T, int D, int MODE>
void launch_lfa_p(LfaParams& p, bool small, hipStream_t st)
{
    constexpr int P = 1024 / D;
    if (small) launch_lfa<T, D, MODE, P / 2>(p, st);
    else launch_lfa<T, D, MODE, P>(p, st);
}

template <typename T>
int lfa_pm_impl(int mode, const float* xyz, const void* nei, int idx_bits, const void* f, int64_t ldf, const float* w1, int64_t ldw1,
                const float* b1, int act1, const void* w2, const float* b2, int act2, const void* wfc, const void* wm, const float* bm,
                int actm, void* out, int64_t ldo, int64_t B, int64_t N, int K, int64_t d, int p_hint, ffb6d_stream_t stream)
{
    constexpr int SZ = El<T>::SZ;
    FFB6D_REQUIRE(mode == 1 || mode == 2, "lfa_pm: mode must be 1 (first half) or 2 (second half)");
    FFB6D_REQUIRE(K == 16, "lfa_pm: K must be 16 (got %d)", K);
    FFB6D_REQUIRE(idx_bits == 32 || idx_bits == 64, "lfa_pm: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(d == 32 || d == 64 || d == 128 || d == 256, "lfa_pm: d must be 32, 64, 128 or 256 (got %lld)", (long long)d);
    FFB6D_REQUIRE(B >= 0 && N >= 0, "lfa_pm: bad shape");
    if (B == 0 || N == 0) return FFB6D_OK;
    const int64_t h = d / 2, cout = mode == 1 ? h : d;
    FFB6D_REQUIRE(xyz && nei && f && w1 && b1 && wfc && wm && bm && out && (mode == 1 || (w2 && b2)), "lfa_pm: null pointer");
    FFB6D_REQUIRE(act1 >= 0 && act1 <= 2 && act2 >= 0 && act2 <= 2 && actm >= 0 && actm <= 2,
                  "lfa_pm: activations must be 0 (none), 1 (relu) or 2 (leaky 0.2)");
    FFB6D_REQUIRE(ldf >= h && ldo >= cout && ldw1 >= 10 && (ldf * SZ) % 16 == 0 && (ldo * SZ) % 16 == 0 &&
                  ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w2) |
                    reinterpret_cast<uintptr_t>(wfc) | reinterpret_cast<uintptr_t>(wm) | reinterpret_cast<uintptr_t>(bm) |
                    reinterpret_cast<uintptr_t>(b2)) & 15) == 0,
                  "lfa_pm: rows must be 16-byte aligned and at least as long as their channel count");
    const int64_t npts = B * N;
    FFB6D_REQUIRE((npts + 64) * ldf * SZ < (1LL << 31) && npts * 16 < (1LL << 31), "lfa_pm: operand larger than the 2 GiB buffer addressing of one launch");
    LfaParams p;
    p.xyz = xyz; p.nei = nei; p.f = f; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.wfc = wfc; p.wm = wm; p.bm = bm; p.out = out;
    p.npts = (int)npts; p.N = (int)N; p.ldf = (int)ldf; p.ldo = (int)ldo; p.ldw1 = (int)ldw1; p.idx64 = idx_bits == 64;
    auto slope = [](int a) { return a == 0 ? 1.f : (a == 1 ? 0.f : 0.2f); };
    p.slope1 = slope(act1); p.slope2 = slope(act2); p.slopem = slope(actm);
    hipStream_t st = as_stream(stream);
    const bool small = p_hint == 2 || (p_hint <= 0 && ffb6d_lfa_pm_small_groups(npts, d));
#define FFB6D_LFA_D(D_)                                                              \
    do {                                                                             \
        if (mode == 1) launch_lfa_p<T, D_, 1>(p, small, st);                         \
        else launch_lfa_p<T, D_, 2>(p, small, st);                                   \
    } while (0)
    switch (d) {
        case 32: FFB6D_LFA_D(32); break;
        case 64: FFB6D_LFA_D(64); break;
        case 128: FFB6D_LFA_D(128); break;
        default: FFB6D_LFA_D(256); break;
    }
#undef FFB6D_LFA_D
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

// 1 = half-size point groups (512 / d points per workgroup instead of 1024 / d): when the full-size groups would leave CUs idle
extern "C" int ffb6d_lfa_pm_small_groups(int64_t npts, int64_t d)
{
    const int64_t P = 1024 / d;
    return P >= 4 && ceil_div(npts, P) < 2 * 256 * 2;
}

extern "C" int ffb6d_lfa_pm(int dtype, int mode, const float* xyz, const void* nei, int idx_bits, const void* f, int64_t ldf,
                            const float* w1, int64_t ldw1, const float* b1, int act1, const void* w2, const float* b2, int act2,
                            const void* wfc, const void* wm, const float* bm, int actm, void* out, int64_t ldo, int64_t B, int64_t N,
                            int K, int64_t d, int p_hint, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dtype == 0 || dtype == 1, "lfa_pm: dtype must be 0 (float32) or 1 (bfloat16)");
    if (dtype == 1)
        return lfa_pm_impl<__bf16>(mode, xyz, nei, idx_bits, f, ldf, w1, ldw1, b1, act1, w2, b2, act2, wfc, wm, bm, actm, out, ldo, B, N, K,
                                   d, p_hint, stream);
    return lfa_pm_impl<float>(mode, xyz, nei, idx_bits, f, ldf, w1, ldw1, b1, act1, w2, b2, act2, wfc, wm, bm, actm, out, ldo, B, N, K, d,
                              p_hint, stream);
}
