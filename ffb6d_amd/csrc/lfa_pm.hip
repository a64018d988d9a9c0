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
//     chunks of one row), the position-encoding half of a pair row is generated in registers (the arithmetic of
//     csrc/posenc_body.h), multiplied with lfa.mlp1 on the matrix cores (K = 10: five fp32 MFMAs, in both precisions) and
//     -- half 2 -- pushed through lfa.mlp2 the same way; both land in one image
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
#include <cstdlib>

#include "common.h"
#include "ffb6d_ops.h"
#include "mfma_pm.h"

namespace ffb6d {
namespace {

using namespace pm;

struct LfaParams {
    const float4* xyz4;   // [B, xfs] rows {x, y, z, -}: point n of frame b at b * xfs + n (xfs >= N: a level may be the prefix of a larger table)
    const void* nei;      // [B * N * 16] int32 / int64 neighbour indices inside the frame
    const void* f;        // [B * N, ldf] point rows of T, first d/2 elements used
    const float* w1;      // [d/2, ldw1] fp32 (BatchNorm folded), columns 0..9 used
    const float* b1;      // [d/2] fp32
    const void* w2;       // MODE 2: [d/2, d/2] of T
    const float* b2;      // MODE 2: [d/2] fp32
    const void* wfc;      // [d, d] of T (no bias)
    const void* wm;       // [d / VL, cout, VL] of T: the output MLP's weight, k-chunked (VL = 16 bytes of consecutive k per channel)
    const float* bm;      // [cout] fp32
    void* out;            // [B * N, ldo] of T
    int npts, N, ldf, ldo, ldw1, idx64, xfs;
    float slope1, slope2, slopem;      // act(v) = max(v, slope * v)
    int n_grp;            // point groups (workgroups with work)
};

// a 16-byte chunk of a row -> its VL values as fp32
template <typename T> __device__ __forceinline__ void unpack_chunk(const u32x4 v, float (&o)[16 / El<T>::SZ])
{
    if constexpr (El<T>::SZ == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __uint_as_float(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[2 * e] = __uint_as_float(v[e] << 16);
            o[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
        }
    }
}

// One GEMM phase of the kernel: acc[TA][TB] += A B over NSTEPS 32-byte steps, fragments through a ring of three register
// stages (the fragments of step s + 2 are requested before step s is multiplied; sched_barrier pins the order -- hipcc
// otherwise sinks the loads below the MFMA groups).  One operand comes from L2 (weights: the same for every group, so their
// first two stages can be requested BEFORE the barrier that publishes the other operand), the other from the LDS images.
// NSTEPS is a compile-time constant: the loop unrolls completely and the ring indices are static.
template <int TA, int TB> struct Ring { u32x4 a[3][TA], b[3][TB]; };

template <typename T, int NSTEPS, int TA, int TB, bool A_GLOBAL, typename LA, typename LB>
__device__ __forceinline__ void gemm_early(Ring<TA, TB>& r, LA&& la, LB&& lb)
{
#pragma unroll
    for (int s = 0; s < 2 && s < NSTEPS; ++s) {
        if constexpr (A_GLOBAL) {
#pragma unroll
            for (int i = 0; i < TA; ++i) r.a[s][i] = la(s, i);
        } else {
#pragma unroll
            for (int j = 0; j < TB; ++j) r.b[s][j] = lb(s, j);
        }
    }
}

template <typename T, int NSTEPS, int TA, int TB, bool A_GLOBAL, typename LA, typename LB>
__device__ __forceinline__ void gemm_run(f32x16 (&acc)[TA][TB], Ring<TA, TB>& r, LA&& la, LB&& lb)
{
#pragma unroll
    for (int s = 0; s < 2 && s < NSTEPS; ++s) {           // the operand gemm_early did not fetch
        if constexpr (!A_GLOBAL) {
#pragma unroll
            for (int i = 0; i < TA; ++i) r.a[s][i] = la(s, i);
        } else {
#pragma unroll
            for (int j = 0; j < TB; ++j) r.b[s][j] = lb(s, j);
        }
    }
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        if (s + 2 < NSTEPS) {
#pragma unroll
            for (int i = 0; i < TA; ++i) r.a[(s + 2) % 3][i] = la(s + 2, i);
#pragma unroll
            for (int j = 0; j < TB; ++j) r.b[(s + 2) % 3][j] = lb(s + 2, j);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<T, TA, TB>(acc, r.a[s % 3], r.b[s % 3]);
        __builtin_amdgcn_sched_barrier(0);
    }
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

template <typename T, int D, int MODE, int P, bool WLDS> struct LfaGeom {
    static constexpr int SZ = El<T>::SZ;
    static constexpr int H = D / 2, COUT = MODE == 1 ? H : D;
    static constexpr int RS = D * SZ + 16;            // image row stride: an odd multiple of 16 bytes (conflict-free ds_read_b128)
    static constexpr int RS2 = H * SZ + 16;           // row stride of the W2 image
    static constexpr int ROWS = 16 * P;               // pair rows of a point group
    static constexpr int S_BYTES = ROWS * RS;         // pair image
    static constexpr int PL_BYTES = P * RS;           // pooled rows
    static constexpr int SRC_BYTES = 2 * (ROWS + P) * 4;      // source rows of the pairs + coordinate-table shift of the points, x 2 groups
    static constexpr int BIAS_BYTES = (2 * H + COUT) * 4;     // b1 | b2 | bm
    // WLDS: the weights live in LDS for the life of the (persistent) workgroup, rows padded to whole 32-row tiles with zeros
    static constexpr int WFC_BYTES = WLDS ? D * RS : 0;
    static constexpr int WM_BYTES = WLDS ? COUT * RS : 0;
    static constexpr int W2_BYTES = WLDS && MODE == 2 ? (H < 32 ? 32 : H) * RS2 : 0;
    static constexpr int LDS = S_BYTES + PL_BYTES + SRC_BYTES + BIAS_BYTES + WFC_BYTES + WM_BYTES + W2_BYTES;
};

// NW = waves per workgroup.  4: the waves share one pair image and split the tiles of every phase (barriers between the phases).
// 1 (d <= 64): a "workgroup" is ONE wave with its own small pair image -- no barrier anywhere, every phase is wave-local, and the
// CU interleaves a dozen fully independent waves; the point groups are a quarter as large.
template <typename T, int D, int MODE, int P, bool WLDS, int NW>
__global__ void __launch_bounds__(64 * NW)
lfa_pm_kernel(const LfaParams p)
{
    constexpr int NT = 64 * NW;                       // threads of the workgroup
    using G = LfaGeom<T, D, MODE, P, WLDS>;
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;                    // k per 32-byte step
    constexpr int H = D / 2;
    constexpr int VL = 16 / SZ;                       // channels per 16-byte chunk
    constexpr int CPR = H / VL;                       // chunks of half a pair row
    constexpr int RS = G::RS, ROWS = G::ROWS;
    constexpr int NRT = P / 2;                        // row tiles (2 points x 16 neighbours)
    constexpr int NCT = D / 32;                       // channel tiles of the score GEMM
    constexpr int NCH = ROWS * CPR / NT;              // chunks per thread of the row gather
    constexpr int RPI = NT / CPR;                     // pair rows covered by one chunk per thread
    constexpr int NI = (ROWS + NT - 1) / NT;          // pair indices per thread
    constexpr int COUT = MODE == 1 ? H : D;
    constexpr int OOB = 0x7ffffff0;
    static_assert(D % 32 == 0 && P % 2 == 0 && P <= 32, "tile geometry");
    static_assert(H % KSTEP == 0, "half a pair row must be whole 32-byte steps");
    static_assert(NT % CPR == 0 && (ROWS * CPR) % NT == 0 && NCH >= 1, "chunks must divide evenly over the threads");
    static_assert(NW == 4 || (NW == 1 && D <= 64 && !WLDS), "one wave per workgroup: d <= 64, weights from L2");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [pair image | pooled rows | source rows x 2 | biases | weights]
    unsigned char* const S = lds;
    unsigned char* const PL = lds + G::S_BYTES;
    int* const SRC = reinterpret_cast<int*>(lds + G::S_BYTES + G::PL_BYTES);                   // [2][ROWS + P]
    float* const B1 = reinterpret_cast<float*>(lds + G::S_BYTES + G::PL_BYTES + G::SRC_BYTES);  // b1 [H] | b2 [H] | bm [COUT]
    float* const B2 = B1 + H;
    float* const BM = B1 + 2 * H;
    unsigned char* const WFC = lds + G::S_BYTES + G::PL_BYTES + G::SRC_BYTES + G::BIAS_BYTES;
    unsigned char* const WM = WFC + G::WFC_BYTES;
    unsigned char* const W2 = WM + G::WM_BYTES;
    constexpr int RS2 = G::RS2, SRCN = ROWS + P;

    // persistent workgroups: XCD x (blockIdx % 8) walks the x-th eighth of the point groups, its workgroups interleaved
    const int per_xcd = (p.n_grp + 7) >> 3;
    const int xcd = blockIdx.x & 7, nwx = gridDim.x >> 3;
    const int g_end = min(p.n_grp, (xcd + 1) * per_xcd);
    int g = xcd * per_xcd + (int)(blockIdx.x >> 3);
    if (g >= g_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // wave -> tiles of the three GEMM phases
    constexpr int CT1 = H / 32 > 0 ? H / 32 : 1;      // mlp1: channel tiles (H = 16: half a tile), row tiles wave, wave + 4, ..
    constexpr int RT1 = (NRT + NW - 1) / NW;
    constexpr int NOT2 = CT1;                         // mlp2: output channel tiles, waves first along them
    constexpr int WR2 = NW / NOT2;
    static_assert(NW % NOT2 == 0 && WR2 >= 1, "mlp2: the waves tile the output channels first");
    constexpr int RT2 = (NRT + WR2 - 1) / WR2;
    constexpr int WC = NCT < NW ? NCT : NW;           // scores: waves along the channel tiles, then along the row tiles
    constexpr int WR = NW / WC;
    constexpr int TN = NCT / WC;
    constexpr int TM = (NRT + WR - 1) / WR;
    const int wc = wave % WC, wr = wave / WC;
    const int ot2 = wave % NOT2, wr2 = wave / NOT2;

    // (the point rows may be a channel slice of a wider row buffer: the descriptor ends with the slice's last element)
    const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(p.f, (((unsigned)p.npts - 1u) * (unsigned)p.ldf + (unsigned)H) * SZ);
    const __amdgpu_buffer_rsrc_t rs_fc = make_rsrc(p.wfc, (unsigned)(D * D * SZ));
    const __amdgpu_buffer_rsrc_t rs_wm = make_rsrc(p.wm, (unsigned)(COUT * D * SZ));
    const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(MODE == 2 ? p.w2 : p.wfc, (unsigned)(H * H * SZ));
    const int fc_vo = (wc * TN * 32 + l31) * D * SZ + 16 * kh;
    const int w2_vo = (ot2 * 32 + l31) * H * SZ + 16 * kh;              // rows past H: out of range -> zeros
    // weight fragments: out of the LDS images (WLDS) or straight from L2
    auto fc_frag = [&](int s, int j) {
        if constexpr (WLDS) return *reinterpret_cast<const u32x4*>(WFC + ((wc * TN + j) * 32 + l31) * RS + s * 32 + 16 * kh);
        else return __builtin_amdgcn_raw_buffer_load_b128(rs_fc, fc_vo + j * 32 * D * SZ + s * 32, 0, 0);
    };
    auto w2_frag = [&](int s, int) {
        if constexpr (WLDS) return *reinterpret_cast<const u32x4*>(W2 + (ot2 * 32 + l31) * RS2 + s * 32 + 16 * kh);
        else return __builtin_amdgcn_raw_buffer_load_b128(rs_w2, w2_vo + s * 32, 0, 0);
    };
    // biases (and, WLDS, the weight images) -> LDS, once; published by the first barrier below
    for (int c = threadIdx.x; c < 2 * H + COUT; c += NT)
        B1[c] = c < H ? p.b1[c] : (c < 2 * H ? (MODE == 2 ? p.b2[c - H] : 0.f) : p.bm[c - 2 * H]);
    if constexpr (WLDS) {
        constexpr int CW = D * SZ / 16, CW2 = H * SZ / 16;              // 16-byte chunks of a weight row
        for (int c = threadIdx.x; c < D * CW; c += NT)
            *reinterpret_cast<u32x4*>(WFC + (c / CW) * RS + (c % CW) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_fc, c * 16, 0, 0);
        for (int c = threadIdx.x; c < COUT * CW; c += NT)              // k-chunked table [k / VL][channel][VL] -> rows of the image
            *reinterpret_cast<u32x4*>(WM + (c % COUT) * RS + (c / COUT) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_wm, c * 16, 0, 0);
        if constexpr (MODE == 2)
            for (int c = threadIdx.x; c < (H < 32 ? 32 : H) * CW2; c += NT)
                *reinterpret_cast<u32x4*>(W2 + (c / CW2) * RS2 + (c % CW2) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_w2, c * 16, 0, 0);
    }

    // lfa.mlp1 stays in registers for the life of the workgroup: W1[channel l31 of tile c][k = 2 t + kh], fp32
    float wa[CT1][5];
#pragma unroll
    for (int c = 0; c < CT1; ++c)
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int ch = c * 32 + l31;
            wa[c][t] = ch < H ? p.w1[(size_t)ch * p.ldw1 + 2 * t + kh] : 0.f;
        }

    // ---------------------------------------------------------------------------------------------------------------
    // Software pipeline over the groups of this workgroup.  What a group reads from HBM is requested one group ahead and
    // waits in registers: while group g is multiplied, the gathered rows and coordinates of group g + nwx are in flight
    // and the neighbour indices of group g + 2 nwx behind them.
    //   indices  -> source rows of the 16 P pairs (frame base + neighbour index; -1 past the last point), one coalesced read
    //               of the index tensor, parked in LDS (two buffers) for the row gather and the position encoding to share;
    //   rows     -> gathered point rows as whole rows (consecutive lanes fetch consecutive 16-byte chunks of one row) and
    //               the coordinates of the pair this lane owns in each of its mlp1 tiles.
    // ---------------------------------------------------------------------------------------------------------------
    const int col = tid % CPR, row0 = tid / CPR;
    const int pp_l = (l31 >> 2) & 1, nb_l = (l31 & 3) + 4 * (l31 >> 3);          // this lane's slot: point of the tile, neighbour
    int lrow[NCH];                                    // image row (slot order) of gathered chunk i
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int row = row0 + i * RPI, pp = row >> 4, nb = row & 15;
        lrow[i] = (pp >> 1) * 32 + ((nb & 3) | ((pp & 1) << 2) | ((nb >> 2) << 3));
    }
    int idxr[NI], xdr[NI];
    u32x4 fch[NCH];
    float4 pq[RT1][2];                                // coordinates of the pair this lane owns in mlp1 tile j: p, q

    auto fetch_idx = [&](int gg) {
        const int n0 = gg * P;
        // a group of P consecutive points touches at most two frames once P <= N
        const int base0 = (n0 / p.N) * p.N, next = base0 + p.N;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int r = tid + k * NT, n = n0 + (r >> 4);
            int v = -1;
            if (gg < g_end && r < ROWS && n < p.npts) {
                const size_t pair = (size_t)n0 * 16 + r;
                const int nbi = p.idx64 ? (int)static_cast<const long long*>(p.nei)[pair] : static_cast<const int*>(p.nei)[pair];
                v = (P <= p.N ? (n < next ? base0 : next) : (n / p.N) * p.N) + nbi;
            }
            idxr[k] = v;
            // shift between a point's row of f (b * N + n) and its row of the coordinate table (b * xfs + n)
            xdr[k] = (P <= p.N ? (n < next ? n0 / p.N : n0 / p.N + 1) : n / p.N) * (p.xfs - p.N);
        }
    };
    auto park_idx = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int r = tid + k * NT;
            if (r < ROWS) {
                SRC[buf * SRCN + r] = idxr[k];
                if ((r & 15) == 0) SRC[buf * SRCN + ROWS + (r >> 4)] = xdr[k];
            }
        }
    };
    auto fetch_rows = [&](int buf, int gg) {
        const int* src = SRC + buf * SRCN;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int sr = src[row0 + i * RPI];
            fch[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, sr >= 0 ? sr * p.ldf * SZ + col * 16 : OOB, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < RT1; ++j) {
            const int rt = wave + NW * j;
            if (rt >= NRT) break;
            const int pp = 2 * rt + pp_l;
            const int sr = src[pp * 16 + nb_l];
            const int xd = src[ROWS + pp];
            // pairs past the last point compute on row 0 -- finite, never stored
            pq[j][0] = p.xyz4[sr >= 0 ? gg * P + pp + xd : 0];
            pq[j][1] = p.xyz4[sr >= 0 ? sr + xd : 0];
        }
    };

    fetch_idx(g);
    park_idx(0);
    __syncthreads();
    fetch_rows(0, g);
    fetch_idx(g + nwx);

    for (int it = 0; g < g_end; g += nwx, ++it) {
        const int n0 = g * P;
        // -----------------------------------------------------------------------------------------------------------
        // 1. position encoding -> lfa.mlp1 on the matrix cores (fp32 in both precisions, K = 10 = five 32x32x2 MFMAs): a
        //    lane computes the encoding of ONE pair (the slot it owns in the tile), channels are the MFMA rows, and it ends
        //    up with 4 x 4 consecutive channels of that pair's row of the image.
        // -----------------------------------------------------------------------------------------------------------
        Ring<1, RT2> ring2;
        if constexpr (MODE == 2) gemm_early<T, H / KSTEP, 1, RT2, true>(ring2, w2_frag, w2_frag);
#pragma unroll
        for (int j = 0; j < RT1; ++j) {
            const int rt = wave + NW * j;
            if (rt >= NRT) break;
            const float px = pq[j][0].x, py = pq[j][0].y, pz = pq[j][0].z, qx = pq[j][1].x, qy = pq[j][1].y, qz = pq[j][1].z;
            const float dx = px - qx, dy = py - qy, dz = pz - qz;
            const float s2 = ((dx * dx) + (dy * dy)) + (dz * dz);       // -ffp-contract=off: no fusion (RandLANet.py:216-223)
            const float dis = sqrtf(s2);
            // encoding [dis, dx, dy, dz, px, py, pz, qx, qy, qz]: half-wave kh supplies k = 2 t + kh
            const float eb[5] = {kh ? dx : dis, kh ? dz : dy, kh ? py : px, kh ? qx : pz, kh ? qz : qy};
            // half 1: this IS the second half of the pair row; half 2: parked in the first half until mlp2 has consumed it
            unsigned char* const srow = S + (rt * 32 + l31) * RS + (MODE == 1 ? H * SZ : 0);
#pragma unroll
            for (int c = 0; c < CT1; ++c) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int t = 0; t < 5; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c][t], eb[t], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = c * 32 + 8 * q + 4 * kh;
                    if (ch >= H) continue;
                    const float4 b4 = *reinterpret_cast<const float4*>(B1 + ch);
                    El<T>::st4(reinterpret_cast<T*>(srow) + ch,
                               make_float4(activate(acc[4 * q] + b4.x, p.slope1), activate(acc[4 * q + 1] + b4.y, p.slope1),
                                           activate(acc[4 * q + 2] + b4.z, p.slope1), activate(acc[4 * q + 3] + b4.w, p.slope1)));
                }
            }
        }

        if constexpr (MODE == 2) {
            // -------------------------------------------------------------------------------------------------------
            // 2. lfa.mlp2: channels = MFMA rows (weights from L2), pair rows = MFMA columns (image, first half) -> second
            //    half of the image
            // -------------------------------------------------------------------------------------------------------
            __syncthreads();
            f32x16 acc[1][RT2];
            zero(acc);
            gemm_run<T, H / KSTEP, 1, RT2, true>(acc, ring2, w2_frag, [&](int s, int j) {
                const int rt = min(wr2 + WR2 * j, NRT - 1);
                return *reinterpret_cast<const u32x4*>(S + (rt * 32 + l31) * RS + s * 32 + 16 * kh);
            });
#pragma unroll
            for (int j = 0; j < RT2; ++j) {
                const int rt = wr2 + WR2 * j;
                if (rt >= NRT) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = ot2 * 32 + 8 * q + 4 * kh;
                    if (ch >= H) continue;
                    const float4 b4 = *reinterpret_cast<const float4*>(B2 + ch);
                    El<T>::st4(reinterpret_cast<T*>(S + (rt * 32 + l31) * RS) + H + ch,
                               make_float4(activate(acc[0][j][4 * q] + b4.x, p.slope2), activate(acc[0][j][4 * q + 1] + b4.y, p.slope2),
                                           activate(acc[0][j][4 * q + 2] + b4.z, p.slope2), activate(acc[0][j][4 * q + 3] + b4.w, p.slope2)));
                }
            }
            __syncthreads();                          // every wave is done reading the parked mlp1 rows
        }
        // gathered point rows -> first half of the image; the next group's source rows -> their LDS buffer
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(S + lrow[i] * RS + col * 16) = fch[i];
        park_idx((it + 1) & 1);
        Ring<TM, TN> ring3;
        gemm_early<T, D / KSTEP, TM, TN, false>(ring3, fc_frag, fc_frag);
        __syncthreads();
        fetch_rows((it + 1) & 1, g + nwx);            // in flight while this group is multiplied
        fetch_idx(g + 2 * nwx);

        // -----------------------------------------------------------------------------------------------------------
        // 3. scores A = S Wfc^T (pair rows = MFMA rows out of the image, channels = MFMA columns, weights from L2), then
        //    in-lane softmax over the 16 neighbours and the weighted sum -> pooled rows
        // -----------------------------------------------------------------------------------------------------------
        {
            f32x16 acc[TM][TN];
            zero(acc);
            gemm_run<T, D / KSTEP, TM, TN, false>(acc, ring3, [&](int s, int i) {
                const int rt = min(wr + WR * i, NRT - 1);
                return *reinterpret_cast<const u32x4*>(S + (rt * 32 + l31) * RS + s * 32 + 16 * kh);
            }, fc_frag);
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
                    const float nm = -m * 1.44269504088896341f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], 1.44269504088896341f, nm));
                        den += e;
                        num = fmaf(fv[r], e, num);
                    }
                    El<T>::st(reinterpret_cast<T*>(PL + (2 * rt + kh) * RS) + ch, num * __builtin_amdgcn_rcpf(den));
                }
            }
        }
        // -----------------------------------------------------------------------------------------------------------
        // 4. output MLP on the pooled rows.
        //    fp32 -- on the vector ALU: the fp32 MFMA runs at the vector FMA rate, so a 32-column tile padded from P <= 32
        //    points only wastes it, and it would sit on one or two waves while the others wait.  Thread = one output channel
        //    x PPT points; its weight row arrives as 16-byte chunks of VL consecutive k (LDS image, or the k-chunked table in
        //    L2: coalesced), the pooled rows are LDS broadcasts.
        //    bf16 -- on the matrix cores (16x the rate, beside the vector ALU): channels = MFMA rows, points = MFMA columns;
        //    the wave that takes tile 0 rotates with the group.  fp32 accumulation either way.
        // -----------------------------------------------------------------------------------------------------------
        if constexpr (SZ == 4) {
            __syncthreads();
            constexpr int TPC = NT / COUT;                             // threads per channel
            constexpr int PPT = (P + TPC - 1) / TPC;                   // points per thread (threads past P * COUT outputs idle)
            static_assert(COUT <= NT && NT % COUT == 0 && PPT >= 1, "output MLP: one channel per thread");
            const int c = tid % COUT, p0 = min((tid / COUT) * PPT, P - 1);
            const bool mine = (tid / COUT) * PPT < P;
            float acc[PPT];
#pragma unroll
            for (int q = 0; q < PPT; ++q) acc[q] = 0.f;
            const unsigned char* xrow = PL + p0 * RS;
#pragma unroll 4
            for (int kc = 0; kc < D / VL; ++kc) {
                u32x4 wv;
                if constexpr (WLDS) wv = *reinterpret_cast<const u32x4*>(WM + c * RS + kc * 16);
                else wv = __builtin_amdgcn_raw_buffer_load_b128(rs_wm, (kc * COUT + c) * 16, 0, 0);
                float w[VL];
                unpack_chunk<T>(wv, w);
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    float x[VL];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(xrow + min(q, P - 1 - p0) * RS + kc * 16), x);
#pragma unroll
                    for (int e = 0; e < VL; ++e) acc[q] = fmaf(w[e], x[e], acc[q]);
                }
            }
            const float bias = BM[c];
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int n = n0 + p0 + q;
                if (mine && p0 + q < P && n < p.npts) El<T>::st(static_cast<T*>(p.out) + (size_t)n * p.ldo + c, activate(acc[q] + bias, p.slopem));
            }
        } else {
            constexpr int NOT = COUT / 32 > 0 ? COUT / 32 : 1;         // channel tiles wo, wo + 4
            constexpr int TMO = (NOT + NW - 1) / NW;
            const int wo = (wave + it) % NW;
            // a fragment = VL consecutive k of one channel = ONE chunk of the k-chunked table; channels past COUT: out of range -> zeros
            auto wm_frag = [&](int s, int i) {
                const int ch = (wo + NW * i) * 32 + l31;
                if constexpr (WLDS) return *reinterpret_cast<const u32x4*>(WM + min(ch, COUT - 1) * RS + s * 32 + 16 * kh);
                else return __builtin_amdgcn_raw_buffer_load_b128(rs_wm, ch < COUT ? ((2 * s + kh) * COUT + ch) * 16 : OOB, 0, 0);
            };
            Ring<TMO, 1> ring4;
            if (wo < NOT) gemm_early<T, D / KSTEP, TMO, 1, true>(ring4, wm_frag, wm_frag);
            __syncthreads();
            if (wo < NOT) {
                f32x16 acc[TMO][1];
                zero(acc);
                const unsigned char* xp = PL + min(l31, P - 1) * RS + 16 * kh;
                gemm_run<T, D / KSTEP, TMO, 1, true>(acc, ring4, wm_frag,
                                                     [&](int s, int) { return *reinterpret_cast<const u32x4*>(xp + s * 32); });
                const int n = n0 + l31;
                if (l31 < P && n < p.npts) {
                    T* orow = static_cast<T*>(p.out) + (size_t)n * p.ldo;
#pragma unroll
                    for (int i = 0; i < TMO; ++i) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ch = (wo + NW * i) * 32 + 8 * q + 4 * kh;
                            if (ch >= COUT) continue;
                            const float4 b4 = *reinterpret_cast<const float4*>(BM + ch);
                            El<T>::st4(orow + ch, make_float4(activate(acc[i][0][4 * q] + b4.x, p.slopem), activate(acc[i][0][4 * q + 1] + b4.y, p.slopem),
                                                              activate(acc[i][0][4 * q + 2] + b4.z, p.slopem), activate(acc[i][0][4 * q + 3] + b4.w, p.slopem)));
                        }
                    }
                }
            }
        }
    }
}

template <typename T, int D, int MODE, int P, bool WLDS, int NW = 4>
void launch_lfa(LfaParams& p, hipStream_t st, int wg_cap)
{
    using G = LfaGeom<T, D, MODE, P, WLDS>;
    p.n_grp = (int)ceil_div(p.npts, P);
    const void* fn = reinterpret_cast<const void*>(&lfa_pm_kernel<T, D, MODE, P, WLDS, NW>);
    // persistent workgroups: as many as are resident at once (registers + LDS), each walks its share of the point groups
    static int per_cu_of[kMaxDevices + 1];                                     // per device (common.h: device_slot)
    const int slot = device_slot();
    int per_cu = cache_get(per_cu_of, slot);
    if (per_cu == 0) {
        int n = 0;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64 * NW, G::LDS) != hipSuccess || n < 1)
            n = 1;
        per_cu = n;
        cache_set(per_cu_of, slot, n);
    }
    const int64_t per_xcd = ceil_div(p.n_grp, 8);                          // groups of one XCD
    // wg_cap (p_hint bits 8..15): cap on the workgroups per XCD (the tests use 1 to make every workgroup walk several groups)
    const int64_t cap = wg_cap > 0 ? (int64_t)wg_cap : (int64_t)1 << 30;
    const unsigned grid = 8u * (unsigned)std::min<int64_t>(std::min<int64_t>(per_xcd, (int64_t)32 * per_cu), cap);
    hipLaunchKernelGGL((lfa_pm_kernel<T, D, MODE, P, WLDS, NW>), dim3(grid), dim3(64 * NW), G::LDS, st, p);
}

// points per workgroup: 16 P d elements of pair image = 64 KB (fp32) whatever the level; `small` halves it (more, smaller
// workgroups).  wlds: fc / mlp / mlp2 weights resident in LDS (d <= 64 only: they must fit beside the pair image).
template <typename T, int D, int MODE>
void launch_lfa_p(LfaParams& p, int size, bool wlds, hipStream_t st, int wg_cap)
{
    constexpr int P = 1024 / D;
    if constexpr (D <= 64) {
        if (wlds) {
            if (size == 3) launch_lfa<T, D, MODE, P / 4, true>(p, st, wg_cap);
            else if (size == 2) launch_lfa<T, D, MODE, P / 2, true>(p, st, wg_cap);
            else launch_lfa<T, D, MODE, P, true>(p, st, wg_cap);
            return;
        }
        if (size == 3) { launch_lfa<T, D, MODE, P / 4, false>(p, st, wg_cap); return; }
        if (size == 4) { launch_lfa<T, D, MODE, 128 / D, false, 1>(p, st, wg_cap); return; }      // one wave per workgroup: 4 / 2 points each
    }
    if (size >= 2) launch_lfa<T, D, MODE, P / 2, false>(p, st, wg_cap);
    else launch_lfa<T, D, MODE, P, false>(p, st, wg_cap);
}

template <typename T>
int lfa_pm_impl(int mode, const float* xyz4, int64_t xfs, const void* nei, int idx_bits, const void* f, int64_t ldf, const float* w1, int64_t ldw1,
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
    FFB6D_REQUIRE(xyz4 && nei && f && w1 && b1 && wfc && wm && bm && out && (mode == 1 || (w2 && b2)), "lfa_pm: null pointer");
    FFB6D_REQUIRE(act1 >= 0 && act1 <= 2 && act2 >= 0 && act2 <= 2 && actm >= 0 && actm <= 2,
                  "lfa_pm: activations must be 0 (none), 1 (relu) or 2 (leaky 0.2)");
    FFB6D_REQUIRE(ldf >= h && ldo >= cout && ldw1 >= 10 && (ldf * SZ) % 16 == 0 && (ldo * SZ) % 16 == 0 &&
                  ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w2) |
                    reinterpret_cast<uintptr_t>(wfc) | reinterpret_cast<uintptr_t>(wm) | reinterpret_cast<uintptr_t>(bm) |
                    reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(b1)) & 15) == 0,
                  "lfa_pm: rows must be 16-byte aligned and at least as long as their channel count");
    const int64_t npts = B * N;
    FFB6D_REQUIRE((npts + 64) * ldf * SZ < (1LL << 31) && npts * 16 < (1LL << 31) && B * xfs < (1LL << 31),
                  "lfa_pm: operand larger than the 2 GiB buffer addressing of one launch");
    FFB6D_REQUIRE(xfs >= N && (reinterpret_cast<uintptr_t>(xyz4) & 15) == 0, "lfa_pm: coordinate table: 16-byte rows, frame stride >= N");
    LfaParams p;
    p.xyz4 = reinterpret_cast<const float4*>(xyz4); p.xfs = (int)xfs; p.nei = nei; p.f = f; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.wfc = wfc; p.wm = wm; p.bm = bm; p.out = out;
    p.npts = (int)npts; p.N = (int)N; p.ldf = (int)ldf; p.ldo = (int)ldo; p.ldw1 = (int)ldw1; p.idx64 = idx_bits == 64;
    auto slope = [](int a) { return a == 0 ? 1.f : (a == 1 ? 0.f : 0.2f); };
    p.slope1 = slope(act1); p.slope2 = slope(act2); p.slopem = slope(actm);
    hipStream_t st = as_stream(stream);
    const int size_hint = p_hint & 7, w_hint = (p_hint >> 3) & 3;
    const int choice = ffb6d_lfa_pm_choice(npts, d, SZ == 2);
    const int size = size_hint ? size_hint : (choice & 7);
    const bool wlds = d <= 64 && size != 4 && (w_hint == 1 || (w_hint == 0 && (choice >> 3) == 1));
#define FFB6D_LFA_D(D_)                                                              \
    do {                                                                             \
        if (mode == 1) launch_lfa_p<T, D_, 1>(p, size, wlds, st, (p_hint >> 8) & 0xff); \
        else launch_lfa_p<T, D_, 2>(p, size, wlds, st, (p_hint >> 8) & 0xff);        \
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

// The p_hint an automatic launch resolves to (pure host logic; measured on the four level shapes of BASELINE configuration 2,
// profiles/r03_lfa_levels.txt).  Point groups: fp32 -- 256 / d points at d = 32 (92 registers: five waves per SIMD), 512 / d at
// d = 64 and 256, 1024 / d at d = 128; bf16 -- 512 / d for d <= 64, else 1024 / d.  Weights: streamed from L2, except bf16 at
// d = 64 (LDS-resident fc / mlp: 25 against 36 us on the first half); in fp32 LDS residence measured within +-3 % at d = 32 and
// 5-20 % slower at d = 64, where the 40 KB cost a resident workgroup.  The one-wave form (size 4) measured within +-5 % of these.
extern "C" int ffb6d_lfa_pm_choice(int64_t npts, int64_t d, int bf16)
{
    (void)npts;
    const int size = bf16 ? (d <= 64 ? 2 : 1) : (d == 32 ? 3 : (d == 128 ? 1 : 2));
    const int w = (bf16 && d == 64) ? 1 : 2;
    return size + 8 * w;
}

extern "C" int ffb6d_lfa_pm(int dtype, int mode, const float* xyz4, int64_t xyz_frame_stride, const void* nei, int idx_bits, const void* f, int64_t ldf,
                            const float* w1, int64_t ldw1, const float* b1, int act1, const void* w2, const float* b2, int act2,
                            const void* wfc, const void* wm, const float* bm, int actm, void* out, int64_t ldo, int64_t B, int64_t N,
                            int K, int64_t d, int p_hint, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dtype == 0 || dtype == 1, "lfa_pm: dtype must be 0 (float32) or 1 (bfloat16)");
    if (dtype == 1)
        return lfa_pm_impl<__bf16>(mode, xyz4, xyz_frame_stride, nei, idx_bits, f, ldf, w1, ldw1, b1, act1, w2, b2, act2, wfc, wm, bm, actm, out, ldo, B, N, K,
                                   d, p_hint, stream);
    return lfa_pm_impl<float>(mode, xyz4, xyz_frame_stride, nei, idx_bits, f, ldf, w1, ldw1, b1, act1, w2, b2, act2, wfc, wm, bm, actm, out, ldo, B, N, K, d,
                              p_hint, stream);
}
