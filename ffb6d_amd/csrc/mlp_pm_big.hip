// ffb6d_amd/csrc/mlp_pm_big.hip -- the long-row shared-MLP GEMMs in bf16 on a 256 x 256 tile with LDS-DMA operand loads (gfx950).
//
// Reference: the same 1x1 Conv + BatchNorm + activation wrappers and fusion plumbing as csrc/mlp_pm.hip
//   (ffb6d/models/pytorch_utils.py:75-129, ffb6d.py:246-253,282-289; cnn/pspnet.py:7-45 for the bottleneck / the folded up-convolution):
//   out[r, m] = act( sum_k W[m, k] * X[r, k] + bias[m] + Y[yrow(r), m] ),   X = [X1 | X2] along k, rows = points / pixels.
//
// Why another form.  In bf16 one v_mfma_f32_32x32x16_bf16 retires 16 k per 16 operand bytes and lane: the 128 x 128 tile of
// mlp_pm_lds_kernel asks the L2s for (128 + 128) x 128 bytes per 16 MFMAs and wave -- two workgroups per CU then pull ~64 bytes
// per cycle and CU, which is what the L2s deliver to 256 CUs at best: 480-700 TFLOP/s (0.19-0.28 of the 2.5 PFLOP/s peak) whatever
// the schedule (profiles/r05_bench_config5_run.json).  A 256 x 256 tile halves the operand bytes per MFMA, and its operand stream
// never touches a register:
//   * 8 waves = 2 (channel halves) x 4 (point quarters); a wave owns 128 channels x 64 points = 4 x 2 MFMA tiles (128 accumulator
//     registers).  One workgroup per CU (128 KB of LDS), two waves per SIMD;
//   * a step = 128 bytes (64 k) of every W and X row of the tile.  The 512 threads request it with eight `buffer_load_dwordx4 ... lds`
//     each (LDS-DMA: global -> LDS without a staging register; hardware range check = zero rows past the end): eight lanes fetch
//     one 128-byte row segment, a wave instruction fills eight consecutive image rows (1 KB, lane-linear as the instruction demands);
//   * image row stride = 128 bytes (no padding is possible with lane-linear destinations), so the 16-byte chunks of a row are
//     stored permuted: LDS chunk c of row r holds the row's chunk c ^ ((r >> 1) & 7).  The permutation is applied on the GLOBAL
//     side (which chunk a lane asks for) and again in the fragment reads; with it the sixteen lanes a ds_read_b128 serves per
//     cycle hit sixteen different 16-byte bank groups (rows r and r + 1 sit 128 bytes apart = the two halves of the 256-byte bank
//     row, (r >> 1) & 7 spreads the eight row pairs of a lane group);
//   * two stages: the requests of step s + 1 are in flight while step s is multiplied (32 MFMAs per wave = >= 2 k cycles of
//     matrix pipe per SIMD); one barrier per step.
// Same products in the same k order per accumulator as the other forms (k ascending in steps of 16 per lane half), same epilogue
// (pm_epilogue) -> bit-identical results to mlp_pm_lds_kernel / mlp_pm_kernel in bf16.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "ffb6d_ops.h"
#include "mlp_pm_common.h"

namespace ffb6d {
namespace pm {
namespace {

constexpr int BIG_BLK = 512;
constexpr int BIG_CB = 128;                           // bytes of every row per step
constexpr int BIG_IMG = 256 * BIG_CB;                 // one operand image: 256 rows x 128 bytes
constexpr int BIG_BIAS = 4 * BIG_IMG;                 // [2 stages][W image | X image] = 128 KB, then the bias of the group's tiles
constexpr int BIG_TPG = 4;                            // at most four tiles in a sequence
constexpr int BIG_LDS = BIG_BIAS + BIG_TPG * 1024;
static_assert(BIG_LDS <= 160 * 1024, "LDS layout");

// A workgroup multiplies p.tpg consecutive channel tiles of ONE point tile (same X rows, the W rows change): the first step of the next
// tile is requested while the last step of the current one is multiplied (its LDS stage is free by then), so a tile's epilogue -- its stores
// drain behind the wave, nothing waits for them -- is followed by MFMAs at once instead of by the workgroup's exit, the launch of the
// next one and a cold first request (~10 us of 36 per K = 1024 tile, profiles/r06_big_gemm_probe_v4.txt).
// VAR (0 = the product form, one tile per workgroup): bit 1 = no epilogue (probe: the stores depend on a condition that never holds);
// bit 2 = the tiles store directly from registers (32-byte pieces; the form of the tile sequences, whose images stay in use);
// 8 = a step's operand requests spread between its MFMA groups instead of one burst in front (rejected); 10 / 14 = timing probes without
// epilogue (wrong results): half of the fragment reads dropped / no operand requests after step 0
template <int VAR>
__global__ void __launch_bounds__(BIG_BLK)
mlp_pm_big_kernel(const PmParams p)
{
    typedef __bf16 T;
    constexpr int SZ = 2;
    constexpr int CB = BIG_CB, IMG = BIG_IMG;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [2 stages][W image | X image], bias of the group's tiles fp32

    // XCD-aware tile order (workgroup t runs on XCD t % 8: observed dispatch rule, speed only): all channel-tile groups of one point
    // tile take consecutive slots of ONE XCD, so a tile's X rows enter that XCD's L2 once
    const int t = blockIdx.x;
    const int n_grp = (p.n_ct + p.tpg - 1) / p.tpg;
    const int xcd = t & 7, sl = t >> 3;
    const int pt = (sl / n_grp) * 8 + xcd;
    const int ct0 = (sl % n_grp) * p.tpg;
    if (pt >= p.n_pt) return;
    const int ntile = min(p.tpg, p.n_ct - ct0);
    const int r0 = pt * 256;

    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int K = p.k1 + p.k2;
    const int kb1 = p.k1 * SZ, kbt = K * SZ;          // row bytes of x1, of [x1 | x2] (= of a W row)
    const int nstage = kbt / CB;                      // whole steps, at least two (launcher)

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.cout * (unsigned)kbt);
    const unsigned x1_rows = p.xidx ? (unsigned)(p.rows / p.P) * (unsigned)p.px : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(p.x1, span_bytes(x1_rows, p.ld1, p.k1, SZ));
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? p.x2 : p.x1, p.x2 ? span_bytes((unsigned)p.rows, p.ld2, p.k2, SZ) : 0u);

    // bias of the group's channels -> LDS (-0.0f where there is none: x + (-0.0f) == x for every x, the sign of a zero included); the
    // epilogue then issues no global load between its stores (a load's wait would also wait for every store before it: one queue)
    float* bias_lds = reinterpret_cast<float*>(lds + BIG_BIAS);
    for (int c = threadIdx.x; c < ntile * 256; c += BIG_BLK) {
        const int ch = ct0 * 256 + c;
        bias_lds[c] = (p.bias && ch < p.cout) ? p.bias[ch] : -0.0f;
    }

    // loader: instruction i (< 4) of wave w fills image rows 64 i + 8 w .. + 7; lane -> row + (lane >> 3), LDS chunk lane & 7, which
    // holds the row's chunk (lane & 7) ^ ((row >> 1) & 7); (row >> 1) & 7 = 4 (w & 1) + (lane >> 4) for every i
    const int lrow = 8 * wave + (lane >> 3);
    const int lchunk = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
    int w_off[4], x1_off[4], x2_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + lrow + 64 * i;
        x1_off[i] = OOB;
        x2_off[i] = OOB;
        if (r < p.rows) {
            int xr = r;
            if (p.xidx)
                xr = (r / p.P) * p.px + (p.idx64 ? (int)static_cast<const long long*>(p.xidx)[r] : static_cast<const int*>(p.xidx)[r]);
            x1_off[i] = xr * p.ld1 * SZ + lchunk;
            x2_off[i] = r * p.ld2 * SZ + lchunk;
        }
    }
    auto set_w = [&](int ct) {                        // the W rows of channel tile ct
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = ct * 256 + lrow + 64 * i;
            w_off[i] = ch < p.cout ? ch * kbt + lchunk : OOB;
        }
    };
    auto request = [&](int s, int stage) {            // step s of the tile w_off points at -> LDS stage: 8 LDS-DMA instructions per thread
        if constexpr ((VAR & 15) == 14) {             // (probe: MFMAs and fragment reads alone)
            if (s > 0) return;
        }
        const int seg = s * CB;
        const bool first = seg < kb1;
        const __amdgpu_buffer_rsrc_t rx = first ? rs_x1 : rs_x2;
        const int xseg = first ? seg : seg - kb1;
        unsigned char* wi = lds + stage * 2 * IMG + wave * (8 * CB);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lds_dma16(rs_w, wi + i * (64 * CB), w_off[i], seg);
            lds_dma16(rx, wi + IMG + i * (64 * CB), first ? x1_off[i] : x2_off[i], xseg);
        }
    };
    auto request_part = [&](int s, int stage, int i) { // quarter i of request(s, stage): rows 64 i .. 64 i + 63 of both images
        if constexpr ((VAR & 15) == 14) return;
        const int seg = s * CB;
        const bool first = seg < kb1;
        const __amdgpu_buffer_rsrc_t rx = first ? rs_x1 : rs_x2;
        unsigned char* wi = lds + stage * 2 * IMG + wave * (8 * CB) + i * (64 * CB);
        lds_dma16(rs_w, wi, w_off[i], seg);
        lds_dma16(rx, wi + IMG, first ? x1_off[i] : x2_off[i], first ? seg : seg - kb1);
    };

    // fragment reads: lane l31 of a 32-row MFMA tile reads chunk 2 ks + kh of its row = LDS chunk (2 ks + kh) ^ ((l31 >> 1) & 7)
    // (tile rows start at multiples of 32); the chunk byte offset of sub-step ks is the one of sub-step 0 XOR 32 ks
    const int fo0 = (kh ^ ((l31 >> 1) & 7)) * 16;
    const int a_row = (wm * 128 + l31) * CB, b_row = IMG + (wn * 64 + l31) * CB;
    f32x16 acc[4][2];
    // one step: 32 MFMAs per wave out of the images of `stage`.  req_s >= 0 (VAR 8, a rejected form): step req_s is requested into the
    // other stage ON THE WAY, a quarter in front of each group of 8 MFMAs, instead of as one burst in front of the step -- measured 3 %
    // slower over the config-5 shapes (profiles/r06_big_gemm_probe_v10.txt): the burst does not hold the MFMAs up, the late quarter does
    auto multiply = [&](int stage, int req_s) {
        const unsigned char* im = lds + stage * 2 * IMG;
        u32x4 wa[2][4], xb[2][2];
        auto frags = [&](int ks, u32x4 (&a)[4], u32x4 (&b)[2]) {
            const int fo = fo0 ^ (ks * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(im + b_row + j * (32 * CB) + fo);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4*>(im + a_row + i * (32 * CB) + fo);
        };
        frags(0, wa[0], xb[0]);
        if constexpr (VAR == 10) frags(1, wa[1], xb[1]);      // (probe: MFMAs and requests without the fragment reads of sub-steps 2, 3)
#pragma unroll
        for (int ks = 0; ks < CB / 32; ++ks) {
            if ((VAR != 10) && ks + 1 < CB / 32) frags(ks + 1, wa[(ks + 1) & 1], xb[(ks + 1) & 1]);
            if (req_s >= 0) request_part(req_s, stage ^ 1, ks);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<T, 4, 2>(acc, wa[ks & 1], xb[ks & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // epilogue geometry of this lane: two output rows (j), their Y rows, as byte offsets into the buffers (dead rows: out of range)
    // (the launcher sends only launches with whole 16-channel groups, 16-byte aligned output rows and 8-byte aligned Y rows: big_form_ok)
    const unsigned y_rows = p.gidx ? (unsigned)(p.rows / p.P) * (unsigned)p.py : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y ? p.y : p.out, p.y ? span_bytes(y_rows, p.ldy, p.cout, SZ) : 0u);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, span_bytes((unsigned)p.rows, p.ldo, p.cout, SZ));
    int o_off[2], y_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = r0 + (wn * 2 + j) * 32 + l31;
        o_off[j] = OOB;
        y_off[j] = OOB;
        if (r < p.rows) {
            o_off[j] = r * p.ldo * SZ;
            if (p.y) {
                int yr = r;
                if (p.gidx)
                    yr = (r / p.P) * p.py + (p.idx64 ? (int)static_cast<const long long*>(p.gidx)[r] : static_cast<const int*>(p.gidx)[r]);
                y_off[j] = yr * p.ldy * SZ;
            }
        }
    }
    const float slope = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : 0.2f);
    const bool hasy = p.y != nullptr;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

    set_w(ct0);
    request(0, 0);
    __syncthreads();                                  // (waits for the LDS-DMA writes of every wave: vmcnt(0) rides in the barrier's fence)
    int g = 0;                                        // steps made: step g lives in LDS stage g & 1
    for (int ti = 0; ti < ntile; ++ti) {
        const int c0 = (ct0 + ti) * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if constexpr (VAR & 16) {
            // The one-tile workgroup's loop with the step boundary moved IN FRONT of a step's last 8 MFMAs: at that point the step's
            // last fragments are in registers (every wave is done with the stage: the request for the step after next may overwrite
            // it) and the next step's images have landed, so the barrier sits there, the next request and the next step's first
            // fragment reads are issued behind it, and the 8 MFMAs run while those fragments travel -- in the plain loop every wave
            // stands behind the barrier with nothing to multiply until 12 fragment reads per wave of all 8 waves have come back.
            u32x4 wa[2][4], xb[2][2];
            auto frags = [&](int stage, int ks, u32x4 (&a)[4], u32x4 (&b)[2]) {
                const unsigned char* im = lds + stage * 2 * IMG;
                const int fo = fo0 ^ (ks * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(im + b_row + j * (32 * CB) + fo);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4*>(im + a_row + i * (32 * CB) + fo);
            };
            request(1, 1);                            // (at least two steps: big_form_ok)
            frags(0, 0, wa[0], xb[0]);
            for (int s = 0; s < nstage; ++s) {
                const int st = s & 1;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    frags(st, ks + 1, wa[(ks + 1) & 1], xb[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step<T, 4, 2>(acc, wa[ks & 1], xb[ks & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s + 1 < nstage) {
                    __syncthreads();                  // (lgkmcnt(0): the last fragments of stage st; vmcnt(0): step s + 1 has landed)
                    if (s + 2 < nstage) request(s + 2, st);
                    frags(st ^ 1, 0, wa[0], xb[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_step<T, 4, 2>(acc, wa[1], xb[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                          // (every wave is done with the images: the epilogue lays the tile over them)
            g = nstage;
        } else {
        for (int s = 0; s + 1 < nstage; ++s) {        // the last step is multiplied below, outside the loop
            const bool carried = s == 0 && ti > 0;    // step 1 was requested before the previous tile's stores (below)
            if constexpr (VAR != 8) {
                if (!carried) request(s + 1, (g + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            multiply(g & 1, (VAR == 8 && !carried) ? s + 1 : -1);
            __builtin_amdgcn_sched_barrier(0);
            if (carried) {
                // the previous tile's 16 stores are younger than the requests of step 1: wait for the requests only.  A plain
                // __syncthreads() would wait for vmcnt(0) -- the stores' drain, ~8 us per K = 1024 tile -- in front of every tile
                lds_dma_wait<16>();
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            ++g;
        }
        if (ti + 1 < ntile) {                         // the next tile's first step: its stage is free, its latency hides under the last multiply
            set_w(ct0 + ti + 1);
            request(0, (g + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        multiply(g & 1, -1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                              // (the next tile's first step has landed, every wave is done with this tile's images)
        ++g;
        }
        if constexpr (VAR & 2) {
            if (p.act != 77) continue;
        }
        // Epilogue: a lane owns one point (l31) and 4 consecutive channels per group; pairs of groups trade halves across the half-waves
        // (v_permlane32_swap) so that every lane stores 16 bytes (as pm_epilogue's bf16 path); sixteen stores per lane back to back, no
        // load between them (a load's wait would also wait for every store before it: one queue) -- the rows of Y the epilogue adds
        // (gathered rows of the p2r fusion, the pyramid prior: 32 eight-byte loads per lane, a lane's 4 channels of group (i, g)) are
        // all requested first; the fragment registers are free by now.  Same arithmetic in the same order as pm_epilogue:
        // (acc + bias) + y, activation, one rounding to bf16.
        // (the bias goes in first, out of LDS, BEFORE the requests below: hipcc drains vmcnt in front of any LDS read that follows an
        // LDS-DMA instruction -- it cannot tell that the DMA writes another part of LDS -- which would put the requests' latency here)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + ti * 256 + (wm * 4 + i) * 32 + 8 * gq + 4 * kh);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j][4 * gq] += b4.x; acc[i][j][4 * gq + 1] += b4.y; acc[i][j][4 * gq + 2] += b4.z; acc[i][j][4 * gq + 3] += b4.w;
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        auto epilogue = [&](auto hasy_tag) {
            constexpr bool HASY = decltype(hasy_tag)::value;
            u32x2 ypre[HASY ? 32 : 1];
            if constexpr (HASY) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int ch = c0 + (wm * 4 + i) * 32 + 8 * gq + 4 * kh;
                            ypre[(j * 4 + i) * 4 + gq] =
                                __builtin_amdgcn_raw_buffer_load_b64(rs_y, (ch < p.cout && y_off[j] != OOB) ? y_off[j] + ch * SZ : OOB, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
                use_here(ypre[31][1]);                // (the youngest Y load: all of them have arrived before the requests below are issued)
                __builtin_amdgcn_sched_barrier(0);
            }
            // the next tile's step 1 is requested BEFORE this tile's stores (its step 0 landed with the barrier above): the wait at the
            // end of that step 0 then covers the requests and leaves the stores draining under two steps of MFMAs
            if (ti + 1 < ntile) request(1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            auto group = [&](int i, int j, int gq) -> u32x2 {   // the lane's 4 channels of group gq of MFMA tile (i, j)
                float v[4] = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                if constexpr (HASY) {
                    const u32x2 y = ypre[(j * 4 + i) * 4 + gq];
                    v[0] += __uint_as_float(y[0] << 16); v[1] += __uint_as_float(y[0] & 0xffff0000u);
                    v[2] += __uint_as_float(y[1] << 16); v[3] += __uint_as_float(y[1] & 0xffff0000u);
                }
                const bf16x4 b = {(__bf16)activate(v[0], slope), (__bf16)activate(v[1], slope), (__bf16)activate(v[2], slope),
                                  (__bf16)activate(v[3], slope)};
                return __builtin_bit_cast(u32x2, b);
            };
            // The tile of a one-tile workgroup (the automatic plan) goes out through LDS: the operand images are dead by now and
            // nothing is in flight into them, so every wave lays its 64 points x 128 channels (16 KB, its own eighth of the images)
            // down as rows of 256 bytes -- 16-byte chunk c of point row r at chunk slot c ^ (r & 15): the 32 rows one write instruction
            // touches spread over all banks -- and reads it back row-wise: a store instruction then writes 4 whole rows of 256 bytes
            // (eight full 128-byte lines) where the direct form below writes 32-byte pieces of 32 rows (32 partial lines; measured
            // ~10 bytes per clock and CU, a quarter of the launch).  Same bits to the same addresses.
            // (compile-time choice: with both forms in one kernel hipcc spills 67 registers; VAR 0 is launched with one tile per workgroup)
            constexpr bool through_lds = (VAR & 4) == 0;
            unsigned char* own = lds + wave * 16384;
            auto piece = [&](int i, int j, int gp) -> u32x4 {   // the lane's 16 bytes of MFMA tile (i, j), channel pair group gp
                const u32x2 lo = group(i, j, 2 * gp), hi = group(i, j, 2 * gp + 1);
                const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                // lower lane: [own group 2gp | upper's group 2gp]; upper lane: [lower's group 2gp+1 | own group 2gp+1]
                return u32x4{sx[0], sy[0], sx[1], sy[1]};
            };
            if constexpr (!through_lds) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const int ch16 = c0 + (wm * 4 + i) * 32 + 16 * gp + 8 * kh;
                            __builtin_amdgcn_raw_buffer_store_b128(piece(i, j, gp), rs_o,
                                                                   (ch16 < p.cout && o_off[j] != OOB) ? o_off[j] + ch16 * SZ : OOB, 0, 0);
                            if constexpr (HASY) __builtin_amdgcn_sched_barrier(0);     // (left free, the scheduler computes many groups ahead and spills them)
                        }
                return;
            }
            const int lw = opaque(lane);                  // (the write slots: computed here, not carried through the tile loop)
            const int wbase = (lw & 31) * 256, wkey = lw & 15, wkh = lw >> 5;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        // row j * 32 + l31 (row & 15 = l31 & 15), chunk 4 i + 2 gp + kh
                        *reinterpret_cast<u32x4*>(own + j * 8192 + wbase + (((4 * i + 2 * gp + wkh) ^ wkey) * 16)) = piece(i, j, gp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            // (LDS instructions of one wave execute in order: the reads below see every lane's writes)
            const int ln = opaque(lane);                  // (row offsets of the 16 stores: computed here, not carried through the tile loop)
            const int rr = ln >> 4, pos = ln & 15;
            const int rbase = rr * 256 + pos * 16;
            const int r_first = r0 + wn * 64 + rr;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(own + k * 1024 + rbase);
                const int r = r_first + 4 * k, ch8 = c0 + wm * 128 + ((pos ^ ((4 * k + rr) & 15)) * 8);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (r < p.rows && ch8 < p.cout) ? (r * p.ldo + ch8) * SZ : OOB, 0, 0);
                if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (hasy) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
}

}  // namespace

bool big_form_ok(const PmParams& p)
{
    const int64_t K = (int64_t)p.k1 + p.k2;
    const int64_t y_rows = p.y ? (p.gidx ? (int64_t)(p.rows / p.P) * p.py : p.rows) : 0;
    return p.act >= 0 && p.act <= 2 && (K * 2) % BIG_CB == 0 && ((int64_t)p.k1 * 2) % BIG_CB == 0 && K * 2 >= 2 * BIG_CB &&
           ((int64_t)p.cout + 256) * K * 2 < (1LL << 31) &&
           // the epilogue: whole 16-channel groups in 16-byte stores through a buffer descriptor, Y rows in 8-byte loads
           (p.cout & 15) == 0 && (p.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 &&
           ((int64_t)p.rows + 256) * p.ldo * 2 < (1LL << 31) &&
           (!p.y || ((p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 7) == 0 && (y_rows + 256) * p.ldy * 2 < (1LL << 31)));
}

template <int VAR>
bool launch_big_var(PmParams& p, int tpg, hipStream_t st)
{
    static int attr_set[kMaxDevices + 1];
    const int slot = device_slot();
    if (!cache_get(attr_set, slot)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_pm_big_kernel<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS) !=
            hipSuccess)
            return false;
        cache_set(attr_set, slot, 1);
    }
    p.n_ct = (int)ceil_div(p.cout, 256);
    p.n_pt = (int)ceil_div(p.rows, 256);
    p.tpg = std::max(1, std::min(tpg, std::min(p.n_ct, BIG_TPG)));
    const int64_t n_grp = ceil_div(p.n_ct, p.tpg);
    const unsigned grid = (unsigned)(ceil_div(p.n_pt, 8) * 8 * n_grp);
    hipLaunchKernelGGL(mlp_pm_big_kernel<VAR>, dim3(grid), dim3(BIG_BLK), BIG_LDS, st, p);
    return true;
}

// tiles per workgroup for plan 0.  Measured (profiles/r06_big_gemm_probe_v5.txt, _v6.txt): sequences of 2-4 tiles are never faster than
// single tiles -- 1024 -> 2304 on 76800 rows: 380 / 385 / 380 / 395 us for 1 / 2 / 3 / 4 tiles, 300 / 303 us without any epilogue -- although
// the next tile's requests run ahead of the stores and nothing waits for the stores' drain.  What a tile's epilogue costs (~80 us of that
// launch = 354 MB at ~4.4 TB/s) is not a wait: stores and LDS-DMA requests share the CU's vector-memory path, which the operand stream
// alone keeps ~85 % busy (the same loop with requests and no MFMAs: 186 us of 300; ~30 bytes per clock and CU), so store time adds to the
// launch wherever the stores are issued.  Sequences stay available (tile_hint 9 + 256 * T; tests), the automatic plan is one tile.
int big_form_plan(int64_t, int64_t) { return 1; }

// plan: bits 0-3 = tiles per workgroup (0: big_form_plan), bits 4-8 = variant
bool launch_pm_big_bf16(PmParams& p, hipStream_t st, int plan)
{
    int tpg = plan & 15;
    if (tpg == 0) tpg = big_form_plan(p.rows, p.cout);
    switch ((plan >> 4) & 31) {
        case 0: return tpg == 1 ? launch_big_var<0>(p, tpg, st) : launch_big_var<4>(p, tpg, st);   // (tile sequences keep their images: direct stores)
        case 2: return launch_big_var<2>(p, tpg, st);
        case 4: return launch_big_var<4>(p, tpg, st);
        case 8: return launch_big_var<8>(p, tpg, st);
        case 10: return launch_big_var<10>(p, tpg, st);
        case 14: return launch_big_var<14>(p, tpg, st);
        case 16: return launch_big_var<16>(p, 1, st);
        case 18: return launch_big_var<18>(p, 1, st);
        case 30: return launch_big_var<30>(p, 1, st);
        default: return false;
    }
}

}  // namespace pm
}  // namespace ffb6d
