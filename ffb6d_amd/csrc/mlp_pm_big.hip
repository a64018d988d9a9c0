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
#include "common.h"
#include "ffb6d_ops.h"
#include "mlp_pm_common.h"

namespace ffb6d {
namespace pm {
namespace {

constexpr int BIG_BLK = 512;
constexpr int BIG_CB = 128;                           // bytes of every row per step
constexpr int BIG_IMG = 256 * BIG_CB;                 // one operand image: 256 rows x 128 bytes
constexpr int BIG_ORS = 256 + 16;                     // row stride of a wave's output image (the LDS row form of the epilogue)
constexpr int BIG_BIAS = 8 * 64 * BIG_ORS;            // [2 stages][W image | X image] = 128 KB, reused as 8 output images of 17 KB; then the bias
constexpr int BIG_LDS = BIG_BIAS + 1024;
static_assert(BIG_BIAS >= 4 * BIG_IMG && BIG_LDS <= 160 * 1024, "LDS layout");

// VAR (probes; 0 = the product form): bit 0 = the waves 4-7 (the second wave of every SIMD) issue their requests after their 16th MFMA
// of the step instead of at its top; bit 1 = no epilogue (the stores depend on a condition that never holds); bit 2 = the shared
// pm_epilogue (bias and Y rows loaded between the stores); bit 3 = output rows through LDS (whole 256-byte row segments per store);
// bit 4 = no operand requests inside the loop, bit 5 = no multiplies inside the loop (timing of the other half; results are garbage)
template <int VAR>
__global__ void __launch_bounds__(BIG_BLK)
mlp_pm_big_kernel(const PmParams p)
{
    typedef __bf16 T;
    constexpr int SZ = 2;
    constexpr int CB = BIG_CB, IMG = BIG_IMG;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [2 stages][W image | X image], bias of the tile [256] fp32

    // XCD-aware tile order (workgroup t runs on XCD t % 8: observed dispatch rule, speed only): all channel tiles of one point tile
    // take consecutive slots of ONE XCD, so a tile's X rows enter that XCD's L2 once
    const int t = blockIdx.x;
    const int xcd = t & 7, sl = t >> 3;
    const int pt = (sl / p.n_ct) * 8 + xcd;
    const int ct = sl % p.n_ct;
    if (pt >= p.n_pt) return;
    const int c0 = ct * 256, r0 = pt * 256;

    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int K = p.k1 + p.k2;
    const int kb1 = p.k1 * SZ, kbt = K * SZ;          // row bytes of x1, of [x1 | x2] (= of a W row)
    const int nstage = kbt / CB;                      // whole steps (launcher)

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.cout * (unsigned)kbt);
    const unsigned x1_rows = p.xidx ? (unsigned)(p.rows / p.P) * (unsigned)p.px : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(p.x1, span_bytes(x1_rows, p.ld1, p.k1, SZ));
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? p.x2 : p.x1, p.x2 ? span_bytes((unsigned)p.rows, p.ld2, p.k2, SZ) : 0u);

    // bias of the tile's channels -> LDS (-0.0f where there is none: x + (-0.0f) == x for every x, the sign of a zero included); the
    // epilogue then issues no global load between its stores (a load's wait would also wait for every store before it: one queue)
    float* bias_lds = reinterpret_cast<float*>(lds + BIG_BIAS);
    if (threadIdx.x < 256) {
        const int ch = c0 + threadIdx.x;
        bias_lds[threadIdx.x] = (p.bias && ch < p.cout) ? p.bias[ch] : -0.0f;
    }

    // loader: instruction i (< 4) of wave w fills image rows 64 i + 8 w .. + 7; lane -> row + (lane >> 3), LDS chunk lane & 7, which
    // holds the row's chunk (lane & 7) ^ ((row >> 1) & 7); (row >> 1) & 7 = 4 (w & 1) + (lane >> 4) for every i
    const int lrow = 8 * wave + (lane >> 3);
    const int lchunk = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
    int w_off[4], x1_off[4], x2_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = c0 + lrow + 64 * i, r = r0 + lrow + 64 * i;
        w_off[i] = ch < p.cout ? ch * kbt + lchunk : OOB;
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
    auto request = [&](int s, int stage) {            // step s -> LDS stage: 8 LDS-DMA instructions per thread
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

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: lane l31 of a 32-row MFMA tile reads chunk 2 ks + kh of its row = LDS chunk (2 ks + kh) ^ ((l31 >> 1) & 7)
    // (tile rows start at multiples of 32); the chunk byte offset of sub-step ks is the one of sub-step 0 XOR 32 ks
    const int fo0 = (kh ^ ((l31 >> 1) & 7)) * 16;
    const int a_row = (wm * 128 + l31) * CB, b_row = IMG + (wn * 64 + l31) * CB;
    // one step: multiply stage `stage` (32 MFMAs per wave); mid >= 0: this wave requests step `mid` after its 16th MFMA
    auto multiply = [&](int stage, int mid) {
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
#pragma unroll
        for (int ks = 0; ks < CB / 32; ++ks) {
            if (ks + 1 < CB / 32) frags(ks + 1, wa[(ks + 1) & 1], xb[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<T, 4, 2>(acc, wa[ks & 1], xb[ks & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VAR & 1) {
                if (ks == 1 && mid >= 0) request(mid, stage ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // epilogue geometry of this lane: two output rows (j), their Y rows, as byte offsets into the buffers (dead rows: out of range)
    const bool wide = !(VAR & 4) && (p.cout & 15) == 0 && (p.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 &&
                      (!p.y || ((p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 7) == 0)) &&
                      ((long long)p.rows + 256) * p.ldo * SZ < (1LL << 31);
    const unsigned y_rows = p.gidx ? (unsigned)(p.rows / p.P) * (unsigned)p.py : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y ? p.y : p.out, (p.y && wide) ? span_bytes(y_rows, p.ldy, p.cout, SZ) : 0u);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, wide ? span_bytes((unsigned)p.rows, p.ldo, p.cout, SZ) : 0u);
    int o_off[2], y_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = r0 + (wn * 2 + j) * 32 + l31;
        o_off[j] = OOB;
        y_off[j] = OOB;
        if (r < p.rows && wide) {
            o_off[j] = r * p.ldo * SZ;
            if (p.y) {
                int yr = r;
                if (p.gidx)
                    yr = (r / p.P) * p.py + (p.idx64 ? (int)static_cast<const long long*>(p.gidx)[r] : static_cast<const int*>(p.gidx)[r]);
                y_off[j] = yr * p.ldy * SZ;
            }
        }
    }
    // the rows of Y the epilogue adds (gathered rows of the p2r fusion, the pyramid prior): 32 eight-byte loads per lane (a lane's 4
    // channels of group (i, g)), requested before the last step is multiplied -- its operands are in LDS, nothing else is in flight
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 ypre[32];
    auto yload = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = c0 + (wm * 4 + i) * 32 + 8 * g + 4 * kh;
                    ypre[(j * 4 + i) * 4 + g] = __builtin_amdgcn_raw_buffer_load_b64(rs_y, (ch < p.cout && y_off[j] != OOB) ? y_off[j] + ch * SZ : OOB, 0, 0);
                }
    };

    request(0, 0);
    __syncthreads();                                  // (waits for the LDS-DMA writes of every wave: vmcnt(0) rides in the barrier's fence)
    const bool late = (VAR & 1) && wave >= 4;
    int s = 0;
    for (; s + 1 < nstage; ++s) {                     // the last step is multiplied below, outside the loop
        if constexpr (!(VAR & 16)) {
            if (!late) request(s + 1, (s + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(VAR & 32)) multiply(s & 1, late ? s + 1 : -1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (p.y && wide) yload();                         // (outside the loop: the 64 registers are live from here on only)
    __builtin_amdgcn_sched_barrier(0);
    multiply(s & 1, -1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                  // every wave is done with the images (the LDS row form below reuses them)
    if constexpr (VAR & 2) {
        if (p.act != 77) return;
    }
    if (!wide) {
        pm_epilogue<T, 4, 2, false>(p, acc, c0, r0, wm, wn, l31, kh);
        return;
    }
    // Epilogue: a lane owns one point (l31) and 4 consecutive channels per group; pairs of groups trade halves across the half-waves
    // (v_permlane32_swap) so that every lane stores 16 bytes (as pm_epilogue's bf16 path); sixteen stores per lane back to back, no
    // load between them.  Same arithmetic in the same order as pm_epilogue: (acc + bias) + y, activation, one rounding to bf16.
    const float slope = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : 0.2f);
    const bool hasy = p.y != nullptr;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto group = [&](int i, int j, int g) -> u32x2 {  // the lane's 4 channels of group g of MFMA tile (i, j): bias, Y, activation, bf16
        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + (wm * 4 + i) * 32 + 8 * g + 4 * kh);
        float v[4] = {acc[i][j][4 * g] + b4.x, acc[i][j][4 * g + 1] + b4.y, acc[i][j][4 * g + 2] + b4.z, acc[i][j][4 * g + 3] + b4.w};
        if (hasy) {
            const u32x2 y = ypre[(j * 4 + i) * 4 + g];
            v[0] += __uint_as_float(y[0] << 16); v[1] += __uint_as_float(y[0] & 0xffff0000u);
            v[2] += __uint_as_float(y[1] << 16); v[3] += __uint_as_float(y[1] & 0xffff0000u);
        }
        const bf16x4 b = {(__bf16)activate(v[0], slope), (__bf16)activate(v[1], slope), (__bf16)activate(v[2], slope),
                          (__bf16)activate(v[3], slope)};
        return __builtin_bit_cast(u32x2, b);
    };
    if constexpr (VAR & 8) {
        // rows through LDS: the wave's 64 points x 128 channels (16 KB + padding, in the operand images nobody reads any more) written in
        // fragment order, read back row by row -- a store instruction covers 4 rows x 256 contiguous bytes
        constexpr int ORS = BIG_ORS;
        unsigned char* oi = lds + wave * (64 * ORS);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<u32x2*>(oi + (j * 32 + l31) * ORS + (i * 32 + 8 * g + 4 * kh) * SZ) = group(i, j, g);
        __builtin_amdgcn_sched_barrier(0);             // (a wave reads only what its own lanes wrote: the compiler's lgkmcnt wait orders it)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4), chunk = lane & 15;
            const u32x4 v = *reinterpret_cast<const u32x4*>(oi + row * ORS + chunk * 16);
            const int r = r0 + wn * 64 + row, ch = c0 + wm * 128 + chunk * 8;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (r < p.rows && ch < p.cout) ? r * p.ldo * SZ + ch * SZ : OOB, 0, 0);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const u32x2 lo = group(i, j, 2 * gp), hi = group(i, j, 2 * gp + 1);
                const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                // lower lane: [own group 2gp | upper's group 2gp]; upper lane: [lower's group 2gp+1 | own group 2gp+1]
                const int ch16 = c0 + (wm * 4 + i) * 32 + 16 * gp + 8 * kh;
                const u32x4 ou = {sx[0], sy[0], sx[1], sy[1]};
                __builtin_amdgcn_raw_buffer_store_b128(ou, rs_o, (ch16 < p.cout && o_off[j] != OOB) ? o_off[j] + ch16 * SZ : OOB, 0, 0);
            }
}

}  // namespace

bool big_form_ok(const PmParams& p)
{
    const int64_t K = (int64_t)p.k1 + p.k2;
    return p.act >= 0 && p.act <= 2 && (K * 2) % BIG_CB == 0 && ((int64_t)p.k1 * 2) % BIG_CB == 0 && K * 2 >= 2 * BIG_CB &&
           ((int64_t)p.cout + 256) * K * 2 < (1LL << 31);
}

template <int VAR>
bool launch_big_var(PmParams& p, hipStream_t st)
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
    const unsigned grid = (unsigned)(ceil_div(p.n_pt, 8) * 8 * p.n_ct);
    hipLaunchKernelGGL(mlp_pm_big_kernel<VAR>, dim3(grid), dim3(BIG_BLK), BIG_LDS, st, p);
    return true;
}

bool launch_pm_big_bf16(PmParams& p, hipStream_t st, int var)
{
    switch (var) {
        case 0: return launch_big_var<0>(p, st);
        case 1: return launch_big_var<1>(p, st);
        case 2: return launch_big_var<2>(p, st);
        case 3: return launch_big_var<3>(p, st);
        case 4: return launch_big_var<4>(p, st);
        case 8: return launch_big_var<8>(p, st);
        case 9: return launch_big_var<9>(p, st);
        case 18: return launch_big_var<18>(p, st);
        case 34: return launch_big_var<34>(p, st);
        default: return false;
    }
}

}  // namespace pm
}  // namespace ffb6d
