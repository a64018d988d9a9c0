// ffb6d_amd/csrc/mlp_pm.hip -- shared MLP on POINT-MAJOR / PIXEL-MAJOR activations ("channels last") for gfx950.
//
// Reference: the 1x1 Conv + BatchNorm + activation wrappers
//   ffb6d/models/pytorch_utils.py:75-129, ffb6d/models/RandLA/pytorch_utils.py:35-111
// and the cat / interpolate / residual plumbing around them (ffb6d.py:245-263,273-298,302-312,
// RandLANet.py:179-184).  Same mathematics, different data layout:
//
//   out[r, m] = act( sum_k W[m, k] * X[r, k]  + bias[m]  + Y[yrow(r), m] ),      X = [X1 | X2] along k
//
// with activations stored one ROW per point / pixel (r runs over all points of all frames, the
// C channels of a row are contiguous) and the weight in the layout nn.Conv stores it, [Cout, Cin].
// Why this layout on MI355X: every gather of the hot path (neighbour features, pixel->point max pooling,
// point->pixel interpolation, `choose`) then moves whole contiguous rows of C*4 bytes instead of one 4-byte
// element per 64-byte sector, and BOTH GEMM operands are K-contiguous, which is exactly the register image
// v_mfma_f32_32x32x2_f32 wants:
//
//   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  A lane loads ONE float4 of its row at
//   k0 + 4*(l>>5): the two half-waves hold k0..k0+3 and k0+4..k0+7, and the MFMA of step t (t = 0..3)
//   multiplies the pairs (k0+t, k0+4+t).  The sum over k is the same set of products in a different order
//   (fp32 GEMM parity bar: 1e-5, tests/test_ops_gpu.py), and it makes a lane's operand 16 contiguous bytes.
//
// So there is NO LDS staging and NO barrier in the k-loop: both operands stream global -> VGPR -> MFMA through
// buffer loads (hardware range check = zero fill past the last row, no exec-mask branches), three register
// stages deep, and the waves of a workgroup share rows only through the L1/L2.  A = weight rows (i = output
// channel), B = activation rows (j = point): accumulator register r of lane l is channel (r&3)+8*(r>>2)+4*(l>>5)
// of point l&31, i.e. a lane owns 4 x 4 consecutive channels of ONE point and the epilogue (bias, gathered row of
// Y, activation) is four 16-byte loads and four 16-byte stores per 32 x 32 tile.
//
// bf16 (BASELINE.json configuration 5: mixed precision, fp32 accumulation): the kernels are templated on the element
// type of the activations / weights.  v_mfma_f32_32x32x16_bf16 takes 8 consecutive k per lane -- again one 16-byte load
// (lane l: k0 + 8*(l>>5) .. +7) -- so the byte arithmetic of the operand stream is identical: a step is 32 bytes of
// every row, i.e. 8 k and four MFMAs in fp32, 16 k and ONE MFMA (16x the rate) in bf16.  Bias and all epilogue
// arithmetic stay fp32; stores round to nearest even (v_cvt_pk_bf16_f32).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "ffb6d_ops.h"
#include "mfma_pm.h"
#include "mlp_pm_common.h"

namespace ffb6d {
namespace {

using namespace pm;

// TM x TN MFMA tiles of 32 (channels) x 32 (points) per wave; WM x WN waves per workgroup.
// KSPLIT: the four waves of the workgroup share ONE TM x TN tile and each take a quarter of K (partial sums meet
// in LDS) -- for the deep layers, where a frame has a few hundred points and K up to 1024: without it a handful of
// workgroups would each walk the whole K.
// (Measured and dropped, profiles/r02_mlp_pm_tiles.txt: 128 x 64 / 64 x 128 tiles per wave and stages of 32 k = one whole
// 128-byte line per row and lane pair -- all within +-5 % of this form on the MFMA-bound layers; for the HBM-bound short-K
// layers a variant that issues ALL operand loads of a tile before the first MFMA -- no faster: those launches run at
// t_MFMA + t_HBM instead of max(t_MFMA, t_HBM), i.e. what is missing is overlap ACROSS tiles, a persistent loop.)
template <typename T, int TM, int TN, int WM, int WN, bool KSPLIT>
__global__ void __launch_bounds__(BLK)
mlp_pm_kernel(const PmParams p)
{
    static_assert(WM * WN == 4, "four waves");
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;                      // k per step: 8 (fp32) or 16 (bf16)
    constexpr int BC = 32 * TM * (KSPLIT ? 1 : WM);     // channels per workgroup
    constexpr int BP = 32 * TN * (KSPLIT ? 1 : WN);     // points per workgroup

    // XCD-aware tile order: workgroup t runs on XCD t % 8 (observed dispatch rule; speed only).  All channel tiles of
    // one point tile get consecutive slots of ONE XCD, so the activation rows are fetched into that XCD's L2 once.
    const int t = blockIdx.x;
    const int xcd = t & 7, s = t >> 3;
    const int pt = (s / p.n_ct) * 8 + xcd;
    const int ct = s % p.n_ct;
    if (pt >= p.n_pt) return;
    const int c0 = ct * BC, r0 = pt * BP;

    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = KSPLIT ? 0 : wave / WN, wn = KSPLIT ? 0 : wave % WN;

    const int K = p.k1 + p.k2;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.cout * (unsigned)K * SZ);
    const unsigned x1_rows = p.xidx ? (unsigned)(p.rows / p.P) * (unsigned)p.px : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(p.x1, span_bytes(x1_rows, p.ld1, p.k1, SZ));
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? p.x2 : p.x1, p.x2 ? span_bytes((unsigned)p.rows, p.ld2, p.k2, SZ) : 0u);

    // per-lane byte offsets of this lane's rows (k = 0); rows past the end are out of range -> zeros
    int w_vo[TM], x1_vo[TN], x2_vo[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) w_vo[i] = (c0 + (wm * TM + i) * 32 + l31) * K * SZ + 16 * kh;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = r0 + (wn * TN + j) * 32 + l31;
        int xr = r;
        bool dead = false;      // gathered rows past the end: a constant offset past every buffer (reads zeros), no arithmetic on it
        if (p.xidx) {           // gathered operand rows: the gather IS the operand load
            dead = r >= p.rows;
            if (!dead)
                xr = (r / p.P) * p.px + (p.idx64 ? (int)static_cast<const long long*>(p.xidx)[r]
                                                 : static_cast<const int*>(p.xidx)[r]);
        }
        x1_vo[j] = dead ? 0x7f000000 : (int)((unsigned)xr * (unsigned)(p.ld1 * SZ) + 16u * kh);      // unsigned: rows past the end may wrap
        x2_vo[j] = (int)((unsigned)r * (unsigned)(p.ld2 * SZ) + 16u * kh);
    }

    const int n1 = p.k1 / KSTEP, nsteps = K / KSTEP;      // steps of 32 bytes per row (k1, k2 multiples of KSTEP)
    // this wave's step range
    int s_beg = 0, s_end = nsteps;
    if (KSPLIT) {
        const int per = (nsteps + 3) >> 2;
        s_beg = wave * per;
        s_end = min(nsteps, s_beg + per);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // three register stages; a stage = 32 bytes of every row of the wave's tile, 16 per half-wave.  The k offset travels in
    // the per-lane offset (the scalar offset of a buffer load is not range checked): a surplus prefetch past the last step
    // reads the following elements of the row buffer or, past its end, zeros -- never used either way.
    u32x4 wa0[TM], wa1[TM], wa2[TM], xb0[TN], xb1[TN], xb2[TN];
    auto load = [&](int step, u32x4 (&wa)[TM], u32x4 (&xb)[TN]) {
        const bool second = step >= n1;
        const __amdgpu_buffer_rsrc_t rx = second ? rs_x2 : rs_x1;
        const int wko = step * 32;                                  // bytes
        const int xko = (second ? step - n1 : step) * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i) wa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_vo[i] + wko, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            xb[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (second ? x2_vo[j] : x1_vo[j]) + xko, 0, 0);
    };

    // one basic block per iteration: the loads of step s+2 fly under the MFMAs of steps s and s+1.  sched_barrier pins
    // the issue order (hipcc otherwise sinks the loads below the MFMA groups and every iteration starts by draining them).
#define FFB6D_PIN() __builtin_amdgcn_sched_barrier(0)
    int st = s_beg;
    load(st, wa0, xb0);
    load(st + 1, wa1, xb1);
    for (; st + 3 <= s_end; st += 3) {
        load(st + 2, wa2, xb2);                FFB6D_PIN();
        mfma_step<T, TM, TN>(acc, wa0, xb0);   FFB6D_PIN();
        load(st + 3, wa0, xb0);                FFB6D_PIN();
        mfma_step<T, TM, TN>(acc, wa1, xb1);   FFB6D_PIN();
        load(st + 4, wa1, xb1);                FFB6D_PIN();
        mfma_step<T, TM, TN>(acc, wa2, xb2);   FFB6D_PIN();
    }
#undef FFB6D_PIN
    if (st < s_end) mfma_step<T, TM, TN>(acc, wa0, xb0);
    if (st + 1 < s_end) mfma_step<T, TM, TN>(acc, wa1, xb1);

    if constexpr (KSPLIT) {
        // partial sums of waves 1..3 -> LDS (lane-contiguous), wave 0 adds them and runs the epilogue
        __shared__ float part[3][TM * TN * 16][64];
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[wave - 1][(i * TN + j) * 16 + r][lane] = acc[i][j][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += part[w][(i * TN + j) * 16 + r][lane];
    }

    pm_epilogue<T, TM, TN>(p, acc, c0, r0, wm, wn, l31, kh);
}

// ---------------------------------------------------------------------------------------------------------------
// Stream form of the same GEMM for the HBM-bound layers: short rows (K <= 128 fp32 / 256 bf16) and cout <= 128, where a launch
// moves far more bytes than it has MFMA work (64 -> 64 on 2.4 M pixels: 1.26 GB against 20 GFLOP).  PMC of mlp_pm_kernel on
// that layer (profiles/r02_mlp_pm_stream_pmc.txt): HBM traffic = the algorithmic bytes, MFMA pipe 35 % busy, waves waiting
// 55 % of their cycles, 3.0 TB/s where a plain copy of the same bytes runs at 5.4.  The limiter is the texture-address path:
// an MFMA-fragment-shaped load or store touches 32 rows x 32 bytes per instruction (64 tag look-ups for 1 KB), and every
// wave re-fetches W that way for every 32-byte step.  Here every global access is whole rows:
//   * persistent workgroups (a few per CU) walk the point tiles; W (all cout rows) sits in LDS for the life of the workgroup;
//   * a wave owns 32 points x ALL channels, so X is read exactly once; the tile's rows are fetched as contiguous 16-byte
//     chunks (16 lanes = one 256-byte row), parked in a wave-private LDS image and picked up from there in fragment order
//     (row stride = odd multiple of 16 bytes: conflict-free ds_read_b128); the loads of the next two tiles are in flight
//     (two register sets, loop unrolled by two) while the current one is multiplied and written;
//   * the epilogue (bias, gathered / added Y row, activation, log-softmax -- the arithmetic of pm_epilogue) deposits the
//     result rows in the same LDS image and they leave as whole rows too.
// No barrier after the W copy: each wave reads and writes only its own image.
// ---------------------------------------------------------------------------------------------------------------
// bias_lds: the bias (zero-filled to 32 * TM channels) beside W in LDS; yreg != nullptr: the Y row of this lane's point, fetched by
// the caller BEFORE it requested the tiles after this one -- a load issued here would be the youngest vector-memory operation
// in flight and waiting for it (s_waitcnt vmcnt(0)) would also wait for the prefetched tiles
template <typename T, int TM, bool LSM>
__device__ __forceinline__ void stream_epilogue(const PmParams& p, f32x16 (&acc)[TM][1], unsigned char* img, int os, int r0,
                                                int l31, int kh, const float* bias_lds, const float4 (*yreg)[4])
{
    const T* yb = static_cast<const T*>(p.y);
    const float slope = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : 0.2f);
    const int r = r0 + l31;
    const bool live = r < p.rows;
    const T* yrow = nullptr;
    if (yb && live && !yreg) {
        long long yr = r;
        if (p.gidx) {
            const long long gi = p.idx64 ? static_cast<const long long*>(p.gidx)[r] : (long long)static_cast<const int*>(p.gidx)[r];
            yr = (long long)(r / p.P) * p.py + gi;
        }
        yrow = yb + yr * p.ldy;
    }
    T* orow = reinterpret_cast<T*>(img + l31 * os);
    auto value = [&](int i, int g, int ch) {
        float4 v = make_float4(acc[i][0][4 * g], acc[i][0][4 * g + 1], acc[i][0][4 * g + 2], acc[i][0][4 * g + 3]);
        if (ch < p.cout && live) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + ch);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            if (yreg) {
                const float4 y4 = yreg[i][g];
                v.x += y4.x; v.y += y4.y; v.z += y4.z; v.w += y4.w;
            } else if (yrow) {
                const float4 y4 = El<T>::ld4(yrow + ch);
                v.x += y4.x; v.y += y4.y; v.z += y4.z; v.w += y4.w;
            }
        }
        return v;
    };
    if (!(LSM && p.act == 3)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = i * 32 + 8 * g + 4 * kh;
                const float4 v = value(i, g, ch);
                El<T>::st4(orow + ch, make_float4(activate(v.x, slope), activate(v.y, slope), activate(v.z, slope), activate(v.w, slope)));
            }
    } else {
        // log_softmax over the channels of a point (pspnet.py:108-112 `final`); a point's channels sit in lanes l and l ^ 32
        float4 v[TM][4];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = i * 32 + 8 * g + 4 * kh;
                v[i][g] = value(i, g, ch);
                if (ch < p.cout) m = fmaxf(fmaxf(fmaxf(m, v[i][g].x), fmaxf(v[i][g].y, v[i][g].z)), v[i][g].w);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (i * 32 + 8 * g + 4 * kh < p.cout)
                    sum += (expf(v[i][g].x - m) + expf(v[i][g].y - m)) + (expf(v[i][g].z - m) + expf(v[i][g].w - m));
        sum += __shfl_xor(sum, 32, 64);
        const float lse = m + logf(sum);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                El<T>::st4(orow + i * 32 + 8 * g + 4 * kh,
                           make_float4(v[i][g].x - lse, v[i][g].y - lse, v[i][g].z - lse, v[i][g].w - lse));
    }
}

template <typename T, int TM, int NS, bool LSM, bool TWO, bool HASY>
__global__ void __launch_bounds__(BLK)
mlp_pm_stream_kernel(const PmParams p)
{
    constexpr int SZ = El<T>::SZ;
    constexpr int CRX = 2 * NS;                       // 16-byte chunks of an X / W image row (K is padded with zeros to NS steps)
    constexpr int XS = NS * 32 + 16;                  // X and W image row stride, bytes
    constexpr int CRO = 32 * TM * SZ / 16;            // chunks of an output image row (capacity)
    constexpr int OS = 32 * TM * SZ + 16;             // output image row stride
    constexpr int IMG = 32 * (XS > OS ? XS : OS);     // bytes of a wave's image (X and output rows share it)
    constexpr int NPRE = NS <= 8 ? 2 : 1;             // tiles of loads in flight ahead of the one being multiplied
    constexpr int OOB = 0x7ffffff0;                   // byte offset past every buffer -> loads return zeros, stores are dropped
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 4 wave images, then W [32 * TM][XS]

    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = p.k1 + p.k2;
    const int c1 = p.k1 * SZ / 16, crx = K * SZ / 16, cro = p.cout * SZ / 16;      // chunks: of x1, of a row [x1|x2], of an output row
    unsigned char* img = lds + wave * IMG;
    unsigned char* w_lds = lds + 4 * IMG;
    float* bias_lds = reinterpret_cast<float*>(w_lds + 32 * TM * XS);          // [32 * TM], zeros past cout / without a bias
    for (int c = threadIdx.x; c < 32 * TM; c += BLK) bias_lds[c] = (p.bias && c < p.cout) ? p.bias[c] : 0.f;

    {   // W -> LDS, once: rows past cout and columns past K read as zeros
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.cout * (unsigned)K * SZ);
        for (int u = threadIdx.x; u < 32 * TM * CRX; u += BLK) {
            const int row = u / CRX, col = u % CRX;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, col < crx ? (row * K * SZ) + col * 16 : OOB, 0, 0);
            *reinterpret_cast<u32x4*>(w_lds + row * XS + col * 16) = v;
        }
        __syncthreads();
    }

    const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(p.x1, span_bytes((unsigned)p.rows, p.ld1, p.k1, SZ));
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(TWO ? p.x2 : p.x1, TWO ? span_bytes((unsigned)p.rows, p.ld2, p.k2, SZ) : 0u);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (unsigned)p.rows * (unsigned)p.ldo * SZ);

    // whole rows: load j of a lane is chunk (64 j + lane) % CRX of image row (64 j + lane) / CRX; chunks past K stay zero
    constexpr int RPI = 64 / CRX;                                      // rows per instruction (CRX <= 32)
    const int lrow = lane / CRX, lchunk = lane % CRX;
    const bool from1 = lchunk < c1, from2 = TWO && !from1 && lchunk < crx;
    const int src1 = from1 ? lchunk * 16 : OOB, src2 = from2 ? (lchunk - c1) * 16 : OOB;
    const int step1 = RPI * p.ld1 * SZ, step2 = RPI * p.ld2 * SZ;

    auto gload = [&](int t, u32x4 (&x)[NS]) {
        const int rbase = t * 128 + wave * 32 + lrow;
        // unsigned: tiles prefetched past the last one may wrap (their loads are deselected below)
        const int o1 = (int)((unsigned)rbase * (unsigned)(p.ld1 * SZ)), o2 = (int)((unsigned)rbase * (unsigned)(p.ld2 * SZ));
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const bool live = rbase + j * RPI < p.rows;
            x[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x1, (live && from1) ? o1 + j * step1 + src1 : OOB, 0, 0);
            if constexpr (TWO) {      // second source: a lane takes its chunk from exactly one of the two, the other load returns zeros
                const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, (live && from2) ? o2 + j * step2 + src2 : OOB, 0, 0);
                x[j] |= b;
            }
        }
    };
    auto stage = [&](const u32x4 (&x)[NS]) {          // registers -> this wave's image
#pragma unroll
        for (int j = 0; j < NS; ++j)
            *reinterpret_cast<u32x4*>(img + (lrow + j * RPI) * XS + lchunk * 16) = x[j];
    };
    // HASY: the epilogue's Y rows (gathered or plain) of tile t, one row per lane, 4 x TM chunks of 4 channels
    float4 yreg[TM][4];
    auto yload = [&](int t) {
        if constexpr (HASY) {
            const int r = t * 128 + wave * 32 + l31;
            const T* yrow = nullptr;
            if (r < p.rows) {
                long long yr = r;
                if (p.gidx) {
                    const long long gi = p.idx64 ? static_cast<const long long*>(p.gidx)[r] : (long long)static_cast<const int*>(p.gidx)[r];
                    yr = (long long)(r / p.P) * p.py + gi;
                }
                yrow = static_cast<const T*>(p.y) + yr * p.ldy;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = i * 32 + 8 * g + 4 * kh;
                    yreg[i][g] = (yrow && ch < p.cout) ? El<T>::ld4(yrow + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
    };
    auto compute = [&](int t) {
        f32x16 acc[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        // fragments of step s + 1 are read while step s is multiplied (two register sets; the pins keep hipcc from hoisting
        // every ds_read of the tile above the first MFMA, which costs 4 * NS * (TM + 1) registers)
        u32x4 wa[2][TM], xf[2][1];
        auto frags = [&](int s, u32x4 (&wf)[TM], u32x4 (&xv)[1]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                wf[i] = *reinterpret_cast<const u32x4*>(w_lds + (i * 32 + l31) * XS + (2 * s + kh) * 16);
            xv[0] = *reinterpret_cast<const u32x4*>(img + l31 * XS + (2 * s + kh) * 16);
        };
        frags(0, wa[0], xf[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) frags(s + 1, wa[(s + 1) & 1], xf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<T, TM, 1>(acc, wa[s & 1], xf[s & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int r0 = t * 128 + wave * 32;
        stream_epilogue<T, TM, LSM>(p, acc, img, OS, r0, l31, kh, bias_lds, HASY ? yreg : nullptr);
        // whole result rows out of the image
        constexpr int OPI = 64 / CRO;                 // rows per instruction (CRO <= 32)
        const int orow = lane / CRO, ochunk = lane % CRO;
        const int ob = (r0 + orow) * p.ldo * SZ + ochunk * 16, ostep = OPI * p.ldo * SZ;
#pragma unroll
        for (int j = 0; j < CRO / 2; ++j) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(img + (orow + j * OPI) * OS + ochunk * 16);
            const bool live = r0 + orow + j * OPI < p.rows && ochunk < cro;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, live ? ob + j * ostep : OOB, 0, 0);
        }
    };

#define FFB6D_PIN() __builtin_amdgcn_sched_barrier(0)
    const int stride = gridDim.x;
    int t = blockIdx.x;
    if constexpr (NPRE == 2) {
        u32x4 xa[NS], xb[NS];
        gload(t, xa);
        gload(t + stride, xb);
        while (true) {
            stage(xa);                    FFB6D_PIN();
            yload(t);                     FFB6D_PIN();
            gload(t + 2 * stride, xa);    FFB6D_PIN();
            compute(t);                   FFB6D_PIN();
            t += stride;
            if (t >= p.n_pt) break;
            stage(xb);                    FFB6D_PIN();
            yload(t);                     FFB6D_PIN();
            gload(t + 2 * stride, xb);    FFB6D_PIN();
            compute(t);                   FFB6D_PIN();
            t += stride;
            if (t >= p.n_pt) break;
        }
    } else {
        u32x4 xa[NS];
        gload(t, xa);
        while (true) {
            stage(xa);                    FFB6D_PIN();
            yload(t);                     FFB6D_PIN();
            gload(t + stride, xa);        FFB6D_PIN();
            compute(t);                   FFB6D_PIN();
            t += stride;
            if (t >= p.n_pt) break;
        }
    }
#undef FFB6D_PIN
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled form for the long-row layers in bf16 (K * SZ > 512 bytes or cout > 128: the p2r fusion GEMMs, the PSP bottleneck).
// In bf16 an MFMA retires 16x the k of an fp32 one per operand byte, so mlp_pm_kernel's fragment-shaped operand loads (each
// wave pulls its own W and X fragments through the texture path, 32 rows x 32 bytes per instruction) bound it at ~230 TFLOP/s.
// Here a workgroup (2 x 2 waves, 128 channels x 128 points) stages 128 bytes of every W and X row per step through LDS:
//   * global side: 8 lanes fetch one 128-byte row segment (a whole cache line), every byte once per workgroup;
//   * LDS images [128 rows][128 + 16 bytes] (odd multiple of 16: conflict-free ds_read_b128 in fragment order), two stages:
//     the loads of step s + 1 are in flight while step s is multiplied, one barrier per step;
//   * each wave multiplies its 64 x 64 tile (2 x 2 MFMA tiles) out of the images; epilogue = pm_epilogue.
// Same products and the same k order per accumulator as mlp_pm_kernel -> identical results.
//
// Round 4 built the PERSISTENT form of this kernel three times (workgroups walking tile lists, the operand stream running across tile
// boundaries: static lists per workgroup; lists per XCD handed out with atomics and stealing; the same with a one-load look at all
// counters) and kept none of them -- the records are under profiles/r04_gemm_*_probe_*.txt and profiles/r04_mid_bench_default_*.json:
//   * alone on the chip the static form is 3 % faster in fp32 and 12 % in bf16 (sum over the ten launches of the step: 4108 against
//     4221 us), the dynamic forms 1 % slower (the hand-out costs the short launches: 768 tiles on 512 workgroups);
//   * inside the benchmarked three-stream step every persistent form is SLOWER than this kernel: driver-style roofline 0.584 (static)
//     and 0.615-0.623 (dynamic) against 0.652, frames/s 349-358 against 356: resident workgroups that hold 72 KB of LDS and 250
//     registers for the whole launch get in the way of the side streams' kernels and the other way round, and one tile per
//     workgroup lets the hardware dispatcher do the balancing;
//   * what the tile boundary costs is the EPILOGUE, not the prologue the persistent loop hides: without its stores the same loop runs
//     the ten launches in 3553 us (118-132 TFLOP/s fp32: 0.75-0.84 of the 2.4 GHz peak at the ~2.1 GHz the chip sustains); the
//     epilogue adds output bytes / ~5 TB/s in fp32 and half that rate in bf16 (~100 cycles per store instruction and CU, whatever
//     its width), and the CU's second workgroup hides none of it although it hides about half of an idle wait of the same length.
//     Measured and NOT the cause: the shape of the stores (whole 512-byte rows through the free LDS stage: within 1 % of the
//     fragment-shaped epilogue), the two workgroups of a CU running in step (s_setprio on one of them, or starting it half a tile
//     late: no change; census: blocks b and b + 256 share a CU, profiles/r04_wg_census.txt), all workgroups bursting together (start
//     spread over 8-32 k cycles: no change), the L2 footprint of the tile walk (blocks of 2 .. all channel tiles: no change).
//   What would remove it is an accumulator-to-memory path that does not sit in the in-order vector-memory queue of the waves that
//   multiply (a second accumulator set drained during the next tile needs the 96 staging registers back: LDS-DMA operand loads) --
//   a different kernel, not a variant of this one.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(BLK, 2)            // two workgroups per CU (LDS allows two): at most 256 registers per lane
mlp_pm_lds_kernel(const PmParams p)
{
    constexpr int SZ = El<T>::SZ;
    constexpr int CB = 128;                           // bytes of every row per step
    constexpr int RS = CB + 16;                       // image row stride
    constexpr int IMG = 128 * RS;                     // one operand image
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [2 stages][W image | X image]

    const int t = blockIdx.x;
    const int xcd = t & 7, sl = t >> 3;
    const int pt = (sl / p.n_ct) * 8 + xcd;
    const int ct = sl % p.n_ct;
    if (pt >= p.n_pt) return;
    const int c0 = ct * 128, r0 = pt * 128;

    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int K = p.k1 + p.k2;
    const int kb1 = p.k1 * SZ, kbt = K * SZ;          // row bytes of x1, of [x1 | x2] (= of a W row)
    const int nstage = (kbt + CB - 1) / CB;

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.cout * (unsigned)kbt);
    const unsigned x1_rows = p.xidx ? (unsigned)(p.rows / p.P) * (unsigned)p.px : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(p.x1, span_bytes(x1_rows, p.ld1, p.k1, SZ));
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? p.x2 : p.x1, p.x2 ? span_bytes((unsigned)p.rows, p.ld2, p.k2, SZ) : 0u);

    // loader: thread -> 16-byte chunk lchunk of the 128-byte segment of rows lrow + 32 i (i < 4), for W and for X.  The
    // launcher guarantees K * SZ % 128 == 0 (no row tail) and, with a second source, k1 * SZ % 128 == 0 (a step comes from
    // ONE of the two: wave-uniform descriptor), so the per-thread offsets are loop invariant and the step travels in the scalar
    // offset of the load; rows past the end sit at an out-of-range offset and read zeros.
    const int lchunk = threadIdx.x & 7, lrow = threadIdx.x >> 3;
    int w_off[4], x1_off[4], x2_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = c0 + lrow + 32 * i, r = r0 + lrow + 32 * i;
        w_off[i] = ch < p.cout ? ch * kbt + lchunk * 16 : OOB;
        x1_off[i] = OOB;
        x2_off[i] = OOB;
        if (r < p.rows) {
            int xr = r;
            if (p.xidx)
                xr = (r / p.P) * p.px + (p.idx64 ? (int)static_cast<const long long*>(p.xidx)[r] : static_cast<const int*>(p.xidx)[r]);
            x1_off[i] = xr * p.ld1 * SZ + lchunk * 16;
            x2_off[i] = r * p.ld2 * SZ + lchunk * 16;
        }
    }

    struct Step { u32x4 w[4], x[4]; };                // one step of loads: 8 x 16 bytes per thread
    auto gload = [&](int s, Step& v) {
        if constexpr (SZ == 2) {
            const int seg = s * CB;
            // bf16: branch-free (descriptor and offsets selected; a step past the last one reads at an out-of-range offset =
            // zeros, never used).  hipcc counts outstanding loads only through straight-line code: behind the wave-uniform
            // branches below it parks a step after s_waitcnt vmcnt(0), i.e. it also waits for the steps fetched after it.
            // Harmless in fp32 (a step is ~4 us of MFMA per SIMD, longer than the memory latency), but a bf16 step is ~0.5 us
            // and the prefetch would be one step deep instead of three.
            const bool live = s < nstage, first = seg < kb1;
            const __amdgpu_buffer_rsrc_t rx = first ? rs_x1 : rs_x2;
            const int xseg = first ? seg : seg - kb1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, live ? w_off[i] : OOB, seg, 0);
                v.x[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, live ? (first ? x1_off[i] : x2_off[i]) : OOB, xseg, 0);
            }
            return;
        }
        if (s >= nstage) return;                      // prefetch past the last step
        const int seg = s * CB;
        if (seg < kb1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[i], seg, 0);
                v.x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x1, x1_off[i], seg, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[i], seg, 0);
                v.x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, x2_off[i], seg - kb1, 0);
            }
        }
    };
    auto park = [&](const Step& v, int stage) {
        unsigned char* wi = lds + stage * 2 * IMG;
        unsigned char* xi = wi + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(wi + (lrow + 32 * i) * RS + lchunk * 16) = v.w[i];
            *reinterpret_cast<u32x4*>(xi + (lrow + 32 * i) * RS + lchunk * 16) = v.x[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto multiply = [&](int stage) {
        const unsigned char* wi = lds + stage * 2 * IMG + (wm * 64 + l31) * RS + kh * 16;
        const unsigned char* xi = lds + stage * 2 * IMG + IMG + (wn * 64 + l31) * RS + kh * 16;
        u32x4 wa[2][2], xb[2][2];
        auto frags = [&](int ks, u32x4 (&a)[2], u32x4 (&b)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const u32x4*>(wi + i * 32 * RS + ks * 32);
                b[i] = *reinterpret_cast<const u32x4*>(xi + i * 32 * RS + ks * 32);
            }
        };
        frags(0, wa[0], xb[0]);
#pragma unroll
        for (int ks = 0; ks < CB / 32; ++ks) {
            if (ks + 1 < CB / 32) frags(ks + 1, wa[(ks + 1) & 1], xb[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<T, 2, 2>(acc, wa[ks & 1], xb[ks & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Three register sets: the loads of steps s + 1 .. s + 3 are in flight while step s is multiplied (a bf16 step is ~500
    // MFMA cycles, far less than the memory latency: one step of prefetch leaves the pipe waiting, measured 605 -> see
    // profiles/r02_mlp_pm_lds_ab.txt).  Two LDS stages, one barrier per step: step s + 1 is parked after step s was read.
#define FFB6D_PIN() __builtin_amdgcn_sched_barrier(0)
#define FFB6D_LDS_ITER(FILL, NEXT)                                   \
    if (s + 1 >= nstage) break;         /* the last step is multiplied below, outside the loop */ \
    gload(s + 3, FILL);                 FFB6D_PIN();                 \
    multiply(s & 1);                    FFB6D_PIN();                 \
    park(NEXT, (s + 1) & 1);                                         \
    __syncthreads();                                                 \
    ++s;
    Step va, vb, vc;
    gload(0, va);
    gload(1, vb);
    gload(2, vc);
    park(va, 0);
    __syncthreads();
    int s = 0;
    while (true) {
        FFB6D_LDS_ITER(va, vb)
        FFB6D_LDS_ITER(vb, vc)
        FFB6D_LDS_ITER(vc, va)
    }
#undef FFB6D_LDS_ITER
    // The last step needs no operand registers any more (its images are in LDS, nothing is left to fetch): the three sets are free, and
    // the rows of Y the epilogue adds -- gathered rows of the p2r fusion / decoder: an index load, then 16 row loads per lane -- are
    // requested NOW and arrive under the last 64 MFMAs instead of after them (profiles/r04_gemm_*_probe: the epilogue is exposed in
    // full, and the gathered rows were about half of it on the fusion GEMMs).  Same sums in the same order: identical results.
    typename El<T>::Raw4 ypre[16];
    const bool pre = p.y && (p.cout & 3) == 0 && (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & (4 * SZ - 1)) == 0 &&
                     (p.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & (4 * SZ - 1)) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    if (pre) {
        const T* yb = static_cast<const T*>(p.y);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = r0 + (wn * 2 + j) * 32 + l31;
            long long yr = min(r, p.rows - 1);        // rows past the end: a valid row, fetched and never used
            if (p.gidx) {
                const long long gi = p.idx64 ? static_cast<const long long*>(p.gidx)[yr] : (long long)static_cast<const int*>(p.gidx)[yr];
                yr = (yr / p.P) * p.py + gi;
            }
            const T* yrow = yb + yr * p.ldy;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = min(c0 + (wm * 2 + i) * 32 + 8 * g + 4 * kh, p.cout - 4);
                    ypre[(j * 2 + i) * 4 + g] = El<T>::ld4raw(yrow + ch);
                }
        }
    }
    FFB6D_PIN();
    multiply(s & 1);
    FFB6D_PIN();
#undef FFB6D_PIN
    pm_epilogue<T, 2, 2, false>(p, acc, c0, r0, wm, wn, l31, kh, ypre, pre);
}

// ---------------------------------------------------------------------------------------------------------------
// Tile-SEQUENCE form of the LDS-tiled GEMM (round 5): the answer to what round 4 measured -- the tile boundary's cost is the epilogue
// (output bytes / ~5 TB/s per launch, ~100 cycles per store instruction and CU, hidden by nothing: 10-13 % of the fp32 launches, a quarter
// of the K = 128 / 256 ones), and a wave cannot multiply while it waits in its own in-order vector-memory queue.  Here a workgroup owns
// `tpg` consecutive channel tiles of ONE point tile (same X rows, W rows change) and
//   * the operand stream runs across the tile boundaries (the loader is two steps ahead of the multiply whatever tile that step is in);
//   * the finished tile's accumulators move to a SECOND register set and its epilogue is cut into 16 pieces (one 4-channel group of one
//     32 x 32 MFMA tile: 4 VALU adds, activation, ONE 16-byte store per lane) that ride inside the first four steps of the NEXT tile,
//     one piece per 16 MFMAs -- the stores trickle out between multiplies instead of as a 64-instruction train in front of the
//     workgroup's exit; the rows of Y a piece adds (gathered rows of the p2r fusion / decoder) are requested one step ahead,
//     BEFORE the operand loads of that step, so waiting for them never waits for the operand stream;
//   * register budget for two accumulator sets: ONE staging register set instead of three (a step is 64 fp32 MFMAs per wave = ~4 k
//     cycles of matrix pipe, longer than the memory latency: the loads of step g + 2 are issued at the top of step g and parked into
//     LDS at the top of step g + 1) -- 2 x 64 accumulators + 32 staging + 32 fragments; the bias of the group's channels sits in LDS.
//   Only the LAST tile of a sequence keeps an exposed epilogue; the grid still has >= ~1000 workgroups on every layer the chooser
//   sends here (the persistent forms of round 4 lost inside the three-stream step because 512 resident workgroups cannot be
//   rebalanced; here the hardware dispatcher still hands out several rounds of workgroups).
// Same products in the same k order per accumulator, same epilogue arithmetic -> bit-identical to mlp_pm_lds_kernel / the tile kernels.
// fp32 only (a bf16 step is ~0.5 k cycles: one staging set cannot hide the memory latency there).
// ---------------------------------------------------------------------------------------------------------------
template <bool TWO>
__global__ void __launch_bounds__(BLK, 2)
mlp_pm_seq_kernel(const PmParams p)
{
    typedef float T;
    constexpr int SZ = 4;
    constexpr int CB = 128;                           // bytes of every row per step
    constexpr int RS = CB + 16;                       // image row stride
    constexpr int IMG = 128 * RS;                     // one operand image
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [2 stages][W image | X image], bias of the group [tpg * 128] fp32

    // LIN (round 6, p.lin_wg > 0): the tiles of the point tiles of ONE XCD (pt = xcd, xcd + 8, ...; channel tiles fastest) form one list,
    // cut into p.lin_wg contiguous sequences whose lengths differ by at most one, the longer ones first in dispatch order.  A sequence may
    // run from one point tile into the next (the X rows change, the W rows start over): with whole-point-tile groups the 2400 tiles of
    // 1024 -> 1024 on 38400 rows made 1200 workgroups of 2 on 512 slots = three rounds of two tiles, now 512 sequences of 4-5.
    const bool LIN = p.lin_wg > 0;
    int pt, ct0, ntile, bias_base, n_bias;
    if (LIN) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int n_pt_x = (p.n_pt - xcd + 7) >> 3, L = n_pt_x * p.n_ct;
        const int base = L / p.lin_wg, rem = L - base * p.lin_wg;
        const int start = j * base + min(j, rem);
        ntile = base + (j < rem ? 1 : 0);
        if (j >= p.lin_wg || ntile == 0) return;
        const int pi = start / p.n_ct;
        ct0 = start - pi * p.n_ct;
        pt = xcd + 8 * pi;
        bias_base = 0;
        n_bias = p.n_ct * 128;
    } else {
        int t = blockIdx.x, tpg = p.tpg, pt_lo = 0, pt_hi = p.pt_b;
        if (t >= p.wg_c) { t -= p.wg_c; tpg = 1; pt_lo = p.pt_c; pt_hi = p.n_pt; }
        else if (t >= p.wg_b) { t -= p.wg_b; tpg = p.tpg_b; pt_lo = p.pt_b; pt_hi = p.pt_c; }
        const int n_grp = (p.n_ct + tpg - 1) / tpg;
        const int xcd = t & 7, sl = t >> 3;
        pt = pt_lo + (sl / n_grp) * 8 + xcd;
        const int grp = sl % n_grp;
        if (pt >= pt_hi) return;
        ct0 = grp * tpg;
        ntile = min(tpg, p.n_ct - ct0);
        bias_base = ct0 * 128;
        n_bias = ntile * 128;
    }
    int r0 = pt * 128;                                // rows of the tile being multiplied (LIN: moves on with the sequence)

    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int K = p.k1 + p.k2;
    const int kb1 = p.k1 * SZ, kbt = K * SZ;
    const int nstage = kbt / CB;                      // >= 4 (launcher)
    const int total = ntile * nstage;                 // steps of this workgroup

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)p.cout * (unsigned)kbt);
    const unsigned x1_rows = p.xidx ? (unsigned)(p.rows / p.P) * (unsigned)p.px : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_x1 = make_rsrc(p.x1, span_bytes(x1_rows, p.ld1, p.k1, SZ));
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(TWO ? p.x2 : p.x1, TWO ? span_bytes((unsigned)p.rows, p.ld2, p.k2, SZ) : 0u);
    const unsigned y_rows = p.gidx ? (unsigned)(p.rows / p.P) * (unsigned)p.py : (unsigned)p.rows;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y ? p.y : p.out, p.y ? span_bytes(y_rows, p.ldy, p.cout, SZ) : 0u);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, span_bytes((unsigned)p.rows, p.ldo, p.cout, SZ));

    // bias of the group's channels -> LDS (-0.0f where there is none: x + (-0.0f) == x for every x, the sign of a zero included)
    float* bias_lds = reinterpret_cast<float*>(lds + 4 * IMG);
    for (int c = threadIdx.x; c < n_bias; c += BLK) {
        const int ch = bias_base + c;
        bias_lds[c] = (p.bias && ch < p.cout) ? p.bias[ch] : -0.0f;
    }

    // loader: thread -> 16-byte chunk lchunk of the 128-byte segment of rows lrow + 32 i (i < 4), for W and for X
    const int lchunk = threadIdx.x & 7, lrow = threadIdx.x >> 3;
    int w_off[4], x1_off[4], x2_off[4];
    auto set_x = [&](int rbase) {                     // the loader's X rows: point tile at row rbase
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rbase + lrow + 32 * i;
            x1_off[i] = OOB;
            x2_off[i] = OOB;
            if (r < p.rows) {
                int xr = r;
                if (p.xidx)
                    xr = (r / p.P) * p.px + (p.idx64 ? (int)static_cast<const long long*>(p.xidx)[r] : static_cast<const int*>(p.xidx)[r]);
                x1_off[i] = xr * p.ld1 * SZ + lchunk * 16;
                x2_off[i] = r * p.ld2 * SZ + lchunk * 16;
            }
        }
    };
    set_x(r0);
    int l_r0 = r0;                                    // the loader's point tile (it runs two steps ahead of the multiplies)
    auto set_w = [&](int ct) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = ct * 128 + lrow + 32 * i;
            w_off[i] = ch < p.cout ? ch * kbt + lchunk * 16 : OOB;
        }
    };
    struct Step { u32x4 w[4], x[4]; };
    Step v;
    int l_s = 0, l_ct = ct0;                          // the loader's position: step of channel tile l_ct
    // Branch-free (descriptor and offsets selected; past the last step: out-of-range offsets, nothing is fetched): hipcc counts
    // outstanding loads exactly only through straight-line code, and the epilogue pieces below wait for THEIR loads by count.
    // (max(off, dead): every offset is in [0, OOB], dead is 0 or OOB -- a select on a wave-uniform condition becomes a branch.)
    auto gload = [&](int dead) {                      // the loader's step -> v; then on to the next step (of the next tile after the last step)
        const int seg = l_s * CB;
        const bool first = !TWO || seg < kb1;
        const __amdgpu_buffer_rsrc_t rx = first ? rs_x1 : rs_x2;
        const int xseg = first ? seg : seg - kb1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, max(w_off[i], dead), seg, 0);
            v.x[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, max(first ? x1_off[i] : x2_off[i], dead), xseg, 0);
        }
        if (++l_s == nstage) {
            l_s = 0;
            ++l_ct;
            if (LIN && l_ct == p.n_ct) {              // on to the next point tile of this XCD (the launcher sends no gathered X rows here:
                l_ct = 0;                             // pure arithmetic, no load inside the counted-wait region)
                l_r0 += 8 * 128;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = l_r0 + lrow + 32 * i;
                    x1_off[i] = r < p.rows ? r * p.ld1 * SZ + lchunk * 16 : OOB;
                    x2_off[i] = r < p.rows ? r * p.ld2 * SZ + lchunk * 16 : OOB;
                }
            }
            set_w(l_ct);                              // past the last tile: never used
        }
    };
    auto park = [&](int stage) {
        unsigned char* wi = lds + stage * 2 * IMG;
        unsigned char* xi = wi + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(wi + (lrow + 32 * i) * RS + lchunk * 16) = v.w[i];
            *reinterpret_cast<u32x4*>(xi + (lrow + 32 * i) * RS + lchunk * 16) = v.x[i];
        }
    };

    // epilogue geometry of this lane: two output rows (j), their Y rows, as byte offsets into the buffers
    int o_off[2], y_off[2], po_off[2], py_off[2];      // of the tile being multiplied / of the finished tile whose epilogue is riding
    auto set_rows = [&](int rbase) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = rbase + (wn * 2 + j) * 32 + l31;
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
    };
    set_rows(r0);
    const float slope = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : 0.2f);
    const float y_on = p.y ? 1.f : 0.f;               // (no Y: the registers hold the zeros of an out-of-range load; x + (-0.0f) == x)

    f32x16 acc[2][2], prev[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // piece q of a finished tile (channel base pc0): MFMA tile (i, j), 4-channel group gq
    //   q = 4 * ph + ks  ->  j = q >> 3, i = (q >> 2) & 1, gq = q & 3
    u32x4 yreg[4];                                    // Y rows of the four pieces of the coming drain step
    int pc0 = 0;                                      // channel base of the tile in `prev`
    auto piece_off = [&](int q, int c0t, const int (&row_off)[2]) {          // byte offset of piece q's 16 bytes in a row buffer, OOB if dead
        const int j = q >> 3, i = (q >> 2) & 1, gq = q & 3;
        const int ch = c0t + (wm * 2 + i) * 32 + 8 * gq + 4 * kh;
        return (ch < p.cout && row_off[j] != OOB) ? row_off[j] + ch * SZ : OOB;      // (a dead row / channel: out of range, nothing moves)
    };
    auto yload = [&](int ph, int c0t) {               // pieces 4 ph .. 4 ph + 3 of the tile at c0t
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) yreg[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, piece_off(4 * ph + ks, c0t, py_off), 0, 0);
    };
    auto piece = [&](int ph, int ks) {
        const int q = 4 * ph + ks;
        const int j = q >> 3, i = (q >> 2) & 1, gq = q & 3;
        const int ch = pc0 + (wm * 2 + i) * 32 + 8 * gq + 4 * kh;
        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + (ch - bias_base));
        u32x4 ou;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float u = prev[i][j][4 * gq + c] + (c == 0 ? b4.x : c == 1 ? b4.y : c == 2 ? b4.z : b4.w);
            const float yv = __uint_as_float(yreg[ks][c]);
            u += y_on != 0.f ? yv : -0.0f;
            ou[c] = __float_as_uint(activate(u, slope));
        }
        __builtin_amdgcn_raw_buffer_store_b128(ou, rs_o, piece_off(q, pc0, po_off), 0, 0);
    };

    int g = 0;                                        // step of the workgroup's sequence being multiplied
    // one step: park the loads of step g + 1, request step g + 2, multiply step g (PH >= 0: with four epilogue pieces of the tile before,
    // each after the first half of its sub-step's MFMAs: its wait for the Y rows and its VALU work sit under the second half)
    auto step = [&](auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        park((g + 1) & 1);                            // (after the last step: into a stage nobody reads any more)
        gload(g + 2 < total ? 0 : OOB);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* wi = lds + (g & 1) * 2 * IMG + (wm * 64 + l31) * RS + kh * 16;
        const unsigned char* xi = lds + (g & 1) * 2 * IMG + IMG + (wn * 64 + l31) * RS + kh * 16;
        u32x4 wa[2][2], xb[2][2];
        auto frags = [&](int ks, u32x4 (&a)[2], u32x4 (&b)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const u32x4*>(wi + i * 32 * RS + ks * 32);
                b[i] = *reinterpret_cast<const u32x4*>(xi + i * 32 * RS + ks * 32);
            }
        };
        auto half = [&](int ks, int h) {              // two of the four (k, k + 4) pairs of sub-step ks: 8 MFMAs
#pragma unroll
            for (int tt = 2 * h; tt < 2 * h + 2; ++tt)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wa[ks & 1][i][tt]), __uint_as_float(xb[ks & 1][j][tt]),
                                                                         acc[i][j], 0, 0, 0);
        };
        frags(0, wa[0], xb[0]);
#pragma unroll
        for (int ks = 0; ks < CB / 32; ++ks) {
            if (ks + 1 < CB / 32) frags(ks + 1, wa[(ks + 1) & 1], xb[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            half(ks, 0);
            if constexpr (PH >= 0) __builtin_amdgcn_sched_barrier(0);
            if constexpr (PH >= 0) piece(PH, ks);
            half(ks, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // Y rows of the next drain step -- issued after this step's pieces have used the registers and before the next step's operand loads
        if constexpr (PH >= 0 && PH < 3) yload(PH + 1, pc0);
        __syncthreads();
        ++g;
    };

    set_w(ct0);
    gload(0);
    park(0);
    gload(0);                                         // total >= nstage >= 4
    __syncthreads();
    int c0 = ct0 * 128;
    for (int s = 0; s < nstage; ++s) step(std::integral_constant<int, -1>{});        // first tile: nothing to drain
    for (int ti = 1; ti < ntile; ++ti) {
        // hand the finished tile to the second register set; its epilogue rides in this tile's first four steps
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                prev[i][j] = acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        pc0 = c0;
        c0 += 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) { po_off[j] = o_off[j]; py_off[j] = y_off[j]; }
        if (LIN && c0 == p.n_ct * 128) {              // this tile is the first one of the next point tile
            c0 = 0;
            r0 += 8 * 128;
            set_rows(r0);
        }
        yload(0, pc0);
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        for (int s = 4; s < nstage; ++s) step(std::integral_constant<int, -1>{});
    }
    pm_epilogue<T, 2, 2, false>(p, acc, c0, r0, wm, wn, l31, kh);     // the last tile's epilogue: the only exposed one
}

// ---------------------------------------------------------------------------------------------------------------
// Attentive pooling with the score GEMM fused in (Att_pooling.forward up to the pooled tensor, RandLANet.py:243-248):
//     S[(n,k), :] = [ F[nei[n,k], :] | G[(n,k), :] ]           feature set: gathered point rows | per-pair rows
//     A = S * W_fc^T                                           scores, never written
//     out[n, m] = sum_k S[(n,k), m] * softmax_k(A[(n,k), m])
// Operand roles are swapped with respect to mlp_pm_kernel: the (n,k) PAIRS are the MFMA's i index, the output channels its
// j index, and the 32 row slots of a tile are two points x 16 neighbours in the order
//     slot rho  ->  point (rho >> 2) & 1, neighbour (rho & 3) + 4 * (rho >> 3)
// so that accumulator register r of lane l is neighbour r of point (l >> 5) for channel (l & 31): a lane holds all 16 scores
// of one (point, channel) and the softmax / weighted sum over the neighbourhood is pure in-lane arithmetic -- no shuffles, no
// LDS.  The gather of the neighbour rows IS the operand load (lane rho loads 16 bytes of row nei[n,k]); the feature values
// the scores are multiplied with are re-read channel-contiguous (128-byte segments, L1-resident after the operand loads).
// ---------------------------------------------------------------------------------------------------------------
struct AttParams {
    const void* w;        // [d, d] fc weight, nn.Conv layout, element type T
    const void* f;        // [B * N, ldf] point rows, first c1 elements used
    const void* nei;      // [B * N * 16] int32/int64 neighbour indices inside the frame
    const void* g;        // [B * N * 16, ldg] pair rows, first c2 elements used
    void* out;            // [B * N, ldo]
    int npts, N, c1, c2, ldf, ldg, ldo, idx64;
    int n_pt, n_ct;
};

template <typename T, int TM, int TN>
__global__ void __launch_bounds__(BLK)
att_pool_pm_kernel(const AttParams p)
{
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;
    constexpr int PTS = 2 * TM * 4;                 // points per workgroup (4 waves along the points)
    constexpr int BC = 32 * TN;                     // channels per workgroup
    const int t = blockIdx.x;
    const int xcd = t & 7, s = t >> 3;
    const int pt = (s / p.n_ct) * 8 + xcd;
    const int ct = s % p.n_ct;
    if (pt >= p.n_pt) return;
    const int c0 = ct * BC;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pbase = pt * PTS + wave * (2 * TM);   // first point of this wave
    const int d = p.c1 + p.c2;
    const T* fb = static_cast<const T*>(p.f);
    const T* gb = static_cast<const T*>(p.g);
    T* ob = static_cast<T*>(p.out);

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (unsigned)d * (unsigned)d * SZ);
    const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(p.f, span_bytes((unsigned)p.npts, p.ldf, p.c1, SZ));
    const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(p.g, span_bytes((unsigned)p.npts * 16u, p.ldg, p.c2, SZ));

    // A operand: this lane's pair of every row tile
    const int a_pp = (l31 >> 2) & 1, a_nb = (l31 & 3) + 4 * (l31 >> 3);
    int f_vo[TM], g_vo[TM], w_vo[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int n = pbase + 2 * i + a_pp;
        f_vo[i] = g_vo[i] = 0x7f000000;                                // past every buffer -> zeros (no arithmetic on a sentinel row)
        if (n < p.npts) {
            const size_t pair = (size_t)n * 16 + a_nb;
            const int nb = p.idx64 ? (int)static_cast<const long long*>(p.nei)[pair] : static_cast<const int*>(p.nei)[pair];
            f_vo[i] = ((n / p.N) * p.N + nb) * p.ldf * SZ + 16 * kh;
            g_vo[i] = (int)pair * p.ldg * SZ + 16 * kh;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) w_vo[j] = (c0 + j * 32 + l31) * d * SZ + 16 * kh;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int n1 = p.c1 / KSTEP, nsteps = d / KSTEP;
    u32x4 xa0[TM], xa1[TM], xa2[TM], wb0[TN], wb1[TN], wb2[TN];
    auto load = [&](int step, u32x4 (&xa)[TM], u32x4 (&wb)[TN]) {
        const bool second = step >= n1;
        const __amdgpu_buffer_rsrc_t rx = second ? rs_g : rs_f;
        const int wko = step * 32, xko = (second ? step - n1 : step) * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            xa[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (second ? g_vo[i] : f_vo[i]) + xko, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_vo[j] + wko, 0, 0);
    };
    int st = 0;
    load(0, xa0, wb0);
    load(1, xa1, wb1);
#define FFB6D_PIN() __builtin_amdgcn_sched_barrier(0)
    for (; st + 3 <= nsteps; st += 3) {
        load(st + 2, xa2, wb2);                FFB6D_PIN();
        mfma_step<T, TM, TN>(acc, xa0, wb0);   FFB6D_PIN();
        load(st + 3, xa0, wb0);                FFB6D_PIN();
        mfma_step<T, TM, TN>(acc, xa1, wb1);   FFB6D_PIN();
        load(st + 4, xa1, wb1);                FFB6D_PIN();
        mfma_step<T, TM, TN>(acc, xa2, wb2);   FFB6D_PIN();
    }
#undef FFB6D_PIN
    if (st < nsteps) mfma_step<T, TM, TN>(acc, xa0, wb0);
    if (st + 1 < nsteps) mfma_step<T, TM, TN>(acc, xa1, wb1);

    // epilogue: lane (channel l31 of each column tile, point kh of each row tile) owns 16 scores = one neighbourhood
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int n = pbase + 2 * i + kh;
        if (n >= p.npts) continue;
        const int fbase = (n / p.N) * p.N;
        int nbr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            nbr[r] = p.idx64 ? (int)static_cast<const long long*>(p.nei)[(size_t)n * 16 + r]
                             : static_cast<const int*>(p.nei)[(size_t)n * 16 + r];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ch = c0 + j * 32 + l31;
            if (ch >= d) continue;
            const bool from_f = ch < p.c1;
            const T* src = from_f ? fb + ch : gb + (ch - p.c1);
            float fv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t row = from_f ? (size_t)(fbase + nbr[r]) * p.ldf : ((size_t)n * 16 + r) * p.ldg;
                fv[r] = El<T>::ld(src + row);
            }
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
            El<T>::st(ob + (size_t)n * p.ldo + ch, num * __builtin_amdgcn_rcpf(den));
        }
    }
}

template <typename T, int TM, int TN, int WM, int WN, bool KSPLIT>
void launch_pm(PmParams& p, hipStream_t st)
{
    constexpr int BC = 32 * TM * (KSPLIT ? 1 : WM), BP = 32 * TN * (KSPLIT ? 1 : WN);
    p.n_ct = (int)ceil_div(p.cout, BC);
    p.n_pt = (int)ceil_div(p.rows, BP);
    const unsigned grid = (unsigned)(ceil_div(p.n_pt, 8) * p.n_ct * 8);
    hipLaunchKernelGGL((mlp_pm_kernel<T, TM, TN, WM, WN, KSPLIT>), dim3(grid), dim3(BLK), 0, st, p);
}

template <typename T>
bool launch_lds(PmParams& p, hipStream_t st)
{
    p.n_ct = (int)ceil_div(p.cout, 128);
    p.n_pt = (int)ceil_div(p.rows, 128);
    constexpr size_t lds = 2 * 2 * 128 * (128 + 16);
    static int attr_set[kMaxDevices + 1];                                      // per device (common.h: device_slot)
    const int slot = device_slot();
    if (!cache_get(attr_set, slot)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_pm_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return false;
        cache_set(attr_set, slot, 1);
    }
    const unsigned grid = (unsigned)(ceil_div(p.n_pt, 8) * p.n_ct * 8);
    hipLaunchKernelGGL((mlp_pm_lds_kernel<T>), dim3(grid), dim3(BLK), lds, st, p);
    return true;
}

// plan bits 4-7 == 15 (no region-B length): balanced contiguous sequences per XCD, bits 0-3 = sequences per workgroup slot
inline bool seq_plan_is_lin(int plan) { return ((plan >> 4) & 15) == 15; }

template <bool TWO>
bool launch_seq(PmParams& p, int plan, hipStream_t st)
{
    p.n_ct = (int)ceil_div(p.cout, 128);
    p.n_pt = (int)ceil_div(p.rows, 128);
    static int attr_set[kMaxDevices + 1];                                      // per device (common.h: device_slot)
    const int slot = device_slot();
    if (!cache_get(attr_set, slot)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_pm_seq_kernel<TWO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * 2 * 128 * (128 + 16) + 8 * 128 * (int)sizeof(float)) != hipSuccess)
            return false;
        cache_set(attr_set, slot, 1);
    }
    if (seq_plan_is_lin(plan) && p.n_ct <= 8 && !p.xidx) {
        // balanced contiguous sequences per XCD, `rounds` of them per workgroup slot (two slots per CU, 32 CUs per XCD)
        const int rounds = std::max(1, std::min(plan & 15, 15));
        const int64_t L = ceil_div(p.n_pt, 8) * p.n_ct;                        // tiles of the fullest XCD
        p.lin_wg = (int)std::min<int64_t>(64 * rounds, L);
        if ((plan >> 8) & 255) p.lin_wg = (int)std::min<int64_t>((plan >> 8) & 255, L);     // explicit sequence count per XCD (tests)
        p.tpg = p.tpg_b = 1; p.pt_b = p.pt_c = p.wg_b = p.wg_c = 0;
        const size_t lds = 2 * 2 * 128 * (128 + 16) + (size_t)p.n_ct * 128 * sizeof(float);
        hipLaunchKernelGGL((mlp_pm_seq_kernel<TWO>), dim3((unsigned)(8 * p.lin_wg)), dim3(BLK), lds, st, p);
        return true;
    }
    if (seq_plan_is_lin(plan)) plan = 2;              // (gathered X rows / more than 8 channel tiles: whole-point-tile pairs)
    // plan: bits 0-3 tiles per workgroup in region A, 4-7 in region B, 8-15 tiles of region B / 16, 16-22 tiles of region C / 16
    p.tpg = std::max(1, std::min(plan & 15, std::min(p.n_ct, 8)));
    p.tpg_b = std::max(1, std::min((plan >> 4) & 15, p.tpg));
    const int64_t tiles_b = (int64_t)((plan >> 8) & 255) * 16, tiles_c = (int64_t)((plan >> 16) & 127) * 16;
    // regions are whole multiples of 8 point tiles (one per XCD), taken from the end of the point-tile range
    const int pts_c = (int)std::min<int64_t>(p.n_pt, ceil_div(ceil_div(tiles_c, p.n_ct), 8) * 8);
    const int pts_b = (int)std::min<int64_t>(p.n_pt - pts_c, ceil_div(ceil_div(tiles_b, p.n_ct), 8) * 8);
    p.pt_c = p.n_pt - pts_c;
    p.pt_b = p.pt_c - pts_b;
    if (p.pt_b % 8) p.pt_b -= p.pt_b % 8;                                   // region A ends on a multiple of 8 (B takes the rest)
    if (p.pt_c % 8 && p.pt_c > p.pt_b) p.pt_c -= p.pt_c % 8;                // so does B
    if (p.pt_c < p.pt_b) p.pt_c = p.pt_b;
    const int64_t grp_a = ceil_div(p.n_ct, p.tpg), grp_b = ceil_div(p.n_ct, p.tpg_b);
    p.wg_b = (int)(p.pt_b / 8 * grp_a * 8);
    p.wg_c = p.wg_b + (int)((p.pt_c - p.pt_b) / 8 * grp_b * 8);
    const int64_t grid = p.wg_c + ceil_div(p.n_pt - p.pt_c, 8) * p.n_ct * 8;
    const size_t lds = 2 * 2 * 128 * (128 + 16) + (size_t)p.tpg * 128 * sizeof(float);
    hipLaunchKernelGGL((mlp_pm_seq_kernel<TWO>), dim3((unsigned)grid), dim3(BLK), lds, st, p);
    return true;
}

// what the tile-sequence form needs beyond the LDS-tiled form's conditions: fp32, >= 4 steps per tile, the vector epilogue
// (whole 16-byte channel groups), rows of Y and of the output inside the 2 GiB range of a buffer descriptor
inline bool seq_form_ok(const PmParams& p, int64_t K, int64_t y_rows)
{
    return p.act != 3 && (K * 4) % 128 == 0 && (p.k1 * 4) % 128 == 0 && K * 4 / 128 >= 4 && (p.cout & 3) == 0 && (p.ldo & 3) == 0 &&
           (!p.y || (p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.y)) & 15) == 0 &&
           ((int64_t)p.rows + 256) * p.ldo * 4 < (1LL << 31) && (!p.y || (y_rows + 256) * p.ldy * 4 < (1LL << 31));
}

template <typename T, int TM, int NS, bool LSM, bool TWO, bool HASY = false>
void launch_stream(PmParams& p, hipStream_t st)
{
    constexpr int SZ = El<T>::SZ;
    p.n_ct = 1;
    p.n_pt = (int)ceil_div(p.rows, 128);
    constexpr size_t XS = NS * 32 + 16, OS = 32 * TM * SZ + 16, IMG = 32 * (XS > OS ? XS : OS);       // as in the kernel
    constexpr size_t lds = 4 * IMG + (size_t)32 * TM * XS + (size_t)32 * TM * 4;                        // + the bias
    const void* fn = reinterpret_cast<const void*>(&mlp_pm_stream_kernel<T, TM, NS, LSM, TWO, HASY>);
    // resident workgroups per CU: registers (the X sets in flight + accumulators) and LDS (four wave images + the W copy)
    static int per_cu_of[kMaxDevices + 1];                                     // per device (common.h: device_slot)
    const int slot = device_slot();
    int per_cu = cache_get(per_cu_of, slot);
    if (per_cu == 0) {
        int n = 0;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, BLK, lds) != hipSuccess || n < 1)
            n = 1;
        per_cu = n;
        cache_set(per_cu_of, slot, n);
    }
    const unsigned grid = (unsigned)std::min<int64_t>(p.n_pt, (int64_t)256 * per_cu);
    hipLaunchKernelGGL((mlp_pm_stream_kernel<T, TM, NS, LSM, TWO, HASY>), dim3(grid), dim3(BLK), lds, st, p);
}

template <typename T, int TM, int TN>
void launch_att(AttParams& p, hipStream_t st)
{
    p.n_ct = (int)ceil_div(p.c1 + p.c2, 32 * TN);
    p.n_pt = (int)ceil_div(p.npts, 2 * TM * 4);
    hipLaunchKernelGGL((att_pool_pm_kernel<T, TM, TN>), dim3((unsigned)(ceil_div(p.n_pt, 8) * p.n_ct * 8)), dim3(BLK), 0, st, p);
}

// the stream form handles: cout <= 128, 96 <= K * SZ <= 512 bytes, plain (ungathered) operand rows, 16-byte aligned output
// rows, and log-softmax only for a single source
template <typename T>
bool stream_form_ok(const PmParams& p, int64_t K)
{
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;
    const int64_t ns = K / KSTEP;
    return p.cout <= 128 && ns >= 3 && ns <= 16 && !p.xidx && (p.cout * SZ) % 16 == 0 && (p.ldo * SZ) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0 &&
           (!p.y || ((p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & (4 * SZ - 1)) == 0)) &&
           ((int64_t)p.rows + 256) * p.ldo * SZ < (1LL << 31) && !(p.act == 3 && (p.k2 > 0 || p.cout > 64));
}

template <typename T>
int mlp_pm_impl(const void* w, const float* bias, const void* x1, int64_t k1, int64_t ld1, const void* x1_idx,
                int64_t x1_rows_per_frame, const void* x2, int64_t k2, int64_t ld2, const void* y, int64_t ldy, const void* y_idx,
                int64_t y_rows_per_frame, int idx_bits, int64_t rows_per_frame, void* out, int64_t ldo, int64_t rows, int64_t cout,
                int act, int tile_hint, ffb6d_stream_t stream)
{
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;
    FFB6D_REQUIRE(rows >= 0 && cout >= 1 && k1 >= KSTEP && k2 >= 0, "mlp_pm: bad shape");
    FFB6D_REQUIRE(k1 % KSTEP == 0 && k2 % KSTEP == 0, "mlp_pm: k1 and k2 must be multiples of %d (got %lld, %lld): pad the rows",
                  KSTEP, (long long)k1, (long long)k2);
    FFB6D_REQUIRE(act >= 0 && act <= 3, "mlp_pm: act must be 0 (none), 1 (relu), 2 (leaky 0.2) or 3 (log_softmax over channels)");
    if (rows == 0) return FFB6D_OK;
    FFB6D_REQUIRE(w && x1 && out, "mlp_pm: null pointer");
    FFB6D_REQUIRE((k2 == 0) == (x2 == nullptr), "mlp_pm: x2 and k2 must come together");
    FFB6D_REQUIRE(ld1 >= k1 && (k2 == 0 || ld2 >= k2) && ldo >= cout, "mlp_pm: row stride smaller than the row");
    FFB6D_REQUIRE((ld1 * SZ) % 16 == 0 && (ld2 * SZ) % 16 == 0 &&
                  ((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                  "mlp_pm: operand rows must be 16-byte aligned");
    FFB6D_REQUIRE(!y_idx || y, "mlp_pm: gather indices without rows to gather");
    const bool indexed = y_idx || x1_idx;
    FFB6D_REQUIRE(!indexed || ((idx_bits == 32 || idx_bits == 64) && rows_per_frame >= 1 && rows % rows_per_frame == 0),
                  "mlp_pm: index arrays need idx_bits 32/64 and rows_per_frame dividing rows");
    FFB6D_REQUIRE((!y_idx || y_rows_per_frame >= 1) && (!x1_idx || x1_rows_per_frame >= 1), "mlp_pm: rows per frame of a gathered source");
    FFB6D_REQUIRE(!y || ldy >= cout, "mlp_pm: ldy smaller than cout");
    FFB6D_REQUIRE(act != 3 || (cout <= 64 && (cout & 3) == 0 && (ldo & 3) == 0 &&
                               ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(y)) & (4 * SZ - 1)) == 0 &&
                               (reinterpret_cast<uintptr_t>(bias) & 15) == 0),
                  "mlp_pm: log_softmax epilogue needs cout <= 64, cout %% 4 == 0 and aligned rows");
    const int64_t K = k1 + k2;
    const int64_t x1_rows = x1_idx ? rows / rows_per_frame * x1_rows_per_frame : rows;
    FFB6D_REQUIRE((x1_rows + 256) * ld1 * SZ < (1LL << 31) && (rows + 256) * (ld2 > 0 ? ld2 : 1) * SZ < (1LL << 31) &&
                  (cout + 128) * K * SZ < (1LL << 31), "mlp_pm: operand larger than the 2 GiB buffer addressing of one launch");
    PmParams p;
    p.w = w; p.bias = bias; p.x1 = x1; p.xidx = x1_idx; p.x2 = x2; p.y = y; p.gidx = y_idx; p.out = out;
    p.rows = (int)rows; p.cout = (int)cout; p.k1 = (int)k1; p.k2 = (int)k2; p.ld1 = (int)ld1; p.ld2 = (int)(k2 ? ld2 : 16 / SZ);
    p.ldy = (int)ldy; p.ldo = (int)ldo; p.P = (int)(indexed ? rows_per_frame : rows); p.py = (int)y_rows_per_frame;
    p.px = (int)x1_rows_per_frame; p.act = act;
    p.idx64 = idx_bits == 64;
    hipStream_t st = as_stream(stream);
    int choice = tile_hint & 255, tpg = tile_hint >> 8;
    if (tile_hint <= 0) {
        choice = ffb6d_mlp_pm_choice(rows, cout, k1, k2, act, SZ == 2, x1_idx != nullptr);
        tpg = choice >> 8;
        choice &= 255;
        if (choice == 6 && !stream_form_ok<T>(p, K)) choice = ffb6d_mlp_pm_tile(rows, cout, K, act);      // misaligned rows
        if (choice == 8) {
            const int64_t y_rows = y ? (y_idx ? rows / rows_per_frame * y_rows_per_frame : rows) : 0;
            if (SZ != 4 || !seq_form_ok(p, K, y_rows)) choice = 7;                                        // misaligned rows / huge buffers
        }
        if (choice == 9 && (SZ != 2 || !big_form_ok(p))) choice = 7;
    }
    if (choice == 8 && tpg <= 0) tpg = ffb6d_mlp_pm_seq_plan(rows, cout);
    switch (choice) {
        case 1: launch_pm<T, 2, 2, 2, 2, false>(p, st); break;      // 128 ch x 128 pt
        case 2: launch_pm<T, 2, 2, 1, 4, false>(p, st); break;      // 64 ch x 256 pt
        case 3: launch_pm<T, 1, 2, 1, 4, false>(p, st); break;      // 32 ch x 256 pt
        case 4: launch_pm<T, 1, 1, 2, 2, false>(p, st); break;      // 64 ch x 64 pt
        case 5: launch_pm<T, 2, 1, 2, 2, true>(p, st); break;       // 64 ch x 32 pt, K over the 4 waves
        case 6: {                                                   // stream form: all channels x 32 pt per wave, whole-row traffic
            const int ns = (int)(K / KSTEP);
            FFB6D_REQUIRE(stream_form_ok<T>(p, K), "mlp_pm: the stream form needs cout <= 128, %d <= K <= %d, no operand gather and "
                          "16-byte aligned output rows", 3 * KSTEP, 16 * KSTEP);
            const int tm = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4);
            const bool two = k2 > 0;
#define FFB6D_STREAM_NS(TM_, LSM_, TWO_)                                          \
    do {                                                                          \
        if (ns <= 4) launch_stream<T, TM_, 4, LSM_, TWO_>(p, st);                 \
        else if (ns <= 8) launch_stream<T, TM_, 8, LSM_, TWO_>(p, st);            \
        else launch_stream<T, TM_, 16, LSM_, TWO_>(p, st);                        \
    } while (0)
#define FFB6D_STREAM_Y(TM_)                                                       \
    do {                                                                          \
        if (ns <= 4) launch_stream<T, TM_, 4, false, false, true>(p, st);         \
        else if (ns <= 8) launch_stream<T, TM_, 8, false, false, true>(p, st);    \
        else launch_stream<T, TM_, 16, false, false, true>(p, st);                \
    } while (0)
#define FFB6D_STREAM(TM_)                                                         \
    do {                                                                          \
        if (two) FFB6D_STREAM_NS(TM_, false, true);                               \
        else if (p.y) FFB6D_STREAM_Y(TM_);          /* Y rows fetched ahead of the tile prefetch */ \
        else FFB6D_STREAM_NS(TM_, false, false);                                  \
    } while (0)
            if (act == 3) { if (tm == 1) FFB6D_STREAM_NS(1, true, false); else FFB6D_STREAM_NS(2, true, false); }
            else if (tm == 1) FFB6D_STREAM(1);
            else if (tm == 2) FFB6D_STREAM(2);
            else FFB6D_STREAM(4);
#undef FFB6D_STREAM
#undef FFB6D_STREAM_Y
#undef FFB6D_STREAM_NS
            break;
        }
        case 7:                                                     // LDS-tiled form: 128 ch x 128 pt, whole-row staging
            FFB6D_REQUIRE(act != 3 && (K * SZ) % 128 == 0 && (k1 * SZ) % 128 == 0,
                          "mlp_pm: the LDS-tiled form has no log_softmax epilogue and needs k1 * %d and K * %d to be multiples of 128", SZ, SZ);
            if (!launch_lds<T>(p, st)) return set_error(FFB6D_ERR_HIP, "mlp_pm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            break;
        case 8: {                                                   // tile-sequence form (fp32): tile_hint = 8 + 256 * tiles per workgroup
            if constexpr (SZ == 4) {
                const int64_t y_rows = y ? (y_idx ? rows / rows_per_frame * y_rows_per_frame : rows) : 0;
                FFB6D_REQUIRE(seq_form_ok(p, K, y_rows), "mlp_pm: the tile-sequence form needs fp32 rows of whole 128-byte segments, K >= 128, "
                              "cout %% 4 == 0, 16-byte aligned output / Y rows and no log_softmax");
                const bool ok = k2 > 0 ? launch_seq<true>(p, tpg, st) : launch_seq<false>(p, tpg, st);
                if (!ok) return set_error(FFB6D_ERR_HIP, "mlp_pm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            } else {
                return set_error(FFB6D_ERR_ARG, "mlp_pm: the tile-sequence form (tile_hint 8) is fp32 only");
            }
            break;
        }
        case 9:                                                     // 256 x 256 tile, LDS-DMA operand loads (bf16; csrc/mlp_pm_big.hip)
            if constexpr (SZ == 2) {
                FFB6D_REQUIRE(big_form_ok(p), "mlp_pm: the 256 x 256 form needs bf16 rows of whole 128-byte segments from each source, "
                              "K >= 128 and no log_softmax");
                if (!launch_pm_big_bf16(p, st, tpg)) return set_error(FFB6D_ERR_HIP, "mlp_pm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            } else {
                return set_error(FFB6D_ERR_ARG, "mlp_pm: the 256 x 256 form (tile_hint 9) is bf16 only");
            }
            break;
        default: return set_error(FFB6D_ERR_ARG, "mlp_pm: unknown tile_hint %d", tile_hint);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

template <typename T>
int att_pool_pm_impl(const void* w_fc, const void* f, int64_t c1, int64_t ldf, const void* nei, int idx_bits, const void* g,
                     int64_t c2, int64_t ldg, void* out, int64_t ldo, int64_t B, int64_t N, int K, ffb6d_stream_t stream)
{
    constexpr int SZ = El<T>::SZ;
    constexpr int KSTEP = 32 / SZ;
    FFB6D_REQUIRE(K == 16, "att_pool_pm: K must be 16 (got %d)", K);
    FFB6D_REQUIRE(idx_bits == 32 || idx_bits == 64, "att_pool_pm: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && N >= 0 && c1 >= KSTEP && c2 >= KSTEP && c1 % KSTEP == 0 && c2 % KSTEP == 0,
                  "att_pool_pm: c1 and c2 must be positive multiples of %d", KSTEP);
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(w_fc && f && nei && g && out, "att_pool_pm: null pointer");
    FFB6D_REQUIRE(ldf >= c1 && ldg >= c2 && ldo >= c1 + c2 && (ldf * SZ) % 16 == 0 && (ldg * SZ) % 16 == 0 &&
                  ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(w_fc)) & 15) == 0,
                  "att_pool_pm: rows must be 16-byte aligned and at least as long as their channel count");
    const int64_t npts = B * N, d = c1 + c2;
    FFB6D_REQUIRE((npts + 64) * ldf * SZ < (1LL << 31) && (npts + 64) * 16 * ldg * SZ < (1LL << 31) && (d + 128) * d * SZ < (1LL << 31),
                  "att_pool_pm: operand larger than the 2 GiB buffer addressing of one launch");
    AttParams p;
    p.w = w_fc; p.f = f; p.nei = nei; p.g = g; p.out = out;
    p.npts = (int)npts; p.N = (int)N; p.c1 = (int)c1; p.c2 = (int)c2; p.ldf = (int)ldf; p.ldg = (int)ldg; p.ldo = (int)ldo;
    p.idx64 = idx_bits == 64;
    hipStream_t st = as_stream(stream);
    if (d <= 32) launch_att<T, 4, 1>(p, st);
    else if (d <= 64) launch_att<T, 2, 2>(p, st);
    else launch_att<T, 1, 4>(p, st);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

// tile choice: big tiles while they still give >= ~1.5 workgroups per CU, K split across the waves when even the
// smallest tile leaves CUs idle (deep layers: a few hundred points per frame, K up to 1024)
extern "C" int ffb6d_mlp_pm_tile(int64_t rows, int64_t cout, int64_t K, int act)
{
    if (act == 3) return cout > 32 ? 2 : 3;          // log_softmax: a wave's tile must span all channels
    const int64_t big = ceil_div(rows, 128) * ceil_div(cout, 128);
    const int64_t mid = ceil_div(rows, 256) * ceil_div(cout, 64);
    const int64_t small = ceil_div(rows, 64) * ceil_div(cout, 64);
    if (cout > 32 && mid >= 384) return 2;            // 64 x 256 measured >= 128 x 128 on every large layer (profiles/r02_mlp_pm_tiles.txt)
    if (cout > 64 && big >= 384) return 1;
    if (cout <= 32) return ceil_div(rows, 256) >= 256 ? 3 : 5;
    if (small >= 512 || K < 64) return 4;
    return 5;
}

// Kernel form for tile_hint 0: the stream form (6) on the short-row layers it is faster on (profiles/r02_mlp_pm_stream_ab.txt:
// every single-source shape with 96..512-byte rows and cout <= 128 once there are >= 128 tiles; two-source rows in bf16 only),
// the LDS-tiled form (7) on the big long-row layers, else the tile shape of ffb6d_mlp_pm_tile.
// A/B switch (ffb6d_mlp_pm_set_big_form): 0 = the automatic choice never takes the 256 x 256 bf16 form
static int g_big_form = 1;
extern "C" void ffb6d_mlp_pm_set_big_form(int on) { g_big_form = on; }

extern "C" int ffb6d_mlp_pm_choice(int64_t rows, int64_t cout, int64_t k1, int64_t k2, int act, int bf16, int x1_gathered)
{
    const int64_t K = k1 + k2, row_bytes = K * (bf16 ? 2 : 4);
    const bool stream = cout <= 128 && row_bytes >= 96 && row_bytes <= 512 && !x1_gathered && rows >= 16384 &&
                        (cout * (bf16 ? 2 : 4)) % 16 == 0 && (k2 == 0 || bf16) && !(act == 3 && (k2 > 0 || cout > 64));
    if (stream) return 6;
    // LDS-tiled form: rows of whole 128-byte segments (both sources), enough 128 x 128 tiles to fill the chip; bit-identical to
    // the tile kernels and faster on every such layer measured (profiles/r02_mlp_pm_lds_ab.txt: bf16 2.0-2.4x, fp32 0-10 %)
    // (fp32 with cout <= 64: half of the 128-channel tile would be padding -- the 64 x 256 tile kernel is 1.45x faster there,
    // profiles/r04_tail_gemm_probe.txt; in bf16 the LDS-tiled form still wins)
    const bool lds = act != 3 && row_bytes % 128 == 0 && (k1 * (bf16 ? 2 : 4)) % 128 == 0 &&
                     ceil_div(rows, 128) * ceil_div(cout, 128) >= 256 && (bf16 || cout > 64);
    if (!lds) return ffb6d_mlp_pm_tile(rows, cout, K, act);
    // tile-sequence form (fp32, >= 4 steps of 128 bytes per tile, at least two channel tiles to put in a sequence): 8 + 256 * tiles per workgroup
    const int plan = ffb6d_mlp_pm_seq_plan(rows, cout);
    if (!bf16 && row_bytes >= 512 && (cout & 3) == 0 && (plan & 15) >= 2) return 8 + 256 * plan;
    // bf16: the 256 x 256 tile with LDS-DMA operand loads (csrc/mlp_pm_big.hip) once the launch is MFMA-side (K >= 256) and its
    // 256-wide tiles still fill the chip's 256 workgroup slots a few times over
    if (bf16 && g_big_form && row_bytes >= 512 && cout >= 192 && ceil_div(rows, 256) * ceil_div(cout, 256) >= 512) return 9;
    return 7;
}

// Schedule of the tile-sequence form (see launch_seq for the bit fields): sequence length T of region A, T / 2 in region B, single tiles in
// region C -- the workgroups the dispatcher hands out last are short, so the 512 slots of the chip drain together.
// A/B (ffb6d_mlp_pm_set_seq_lin): 0 = whole-point-tile groups in three regions (round 5) for every layer; 1 = balanced sequences, at least
// two per workgroup slot; 2 = ONE balanced sequence per slot (fastest alone on the chip, profiles/r06_seq_lin_probe.txt)
static int g_seq_lin = 0;
extern "C" void ffb6d_mlp_pm_set_seq_lin(int on) { g_seq_lin = on; }

extern "C" int ffb6d_mlp_pm_seq_plan(int64_t rows, int64_t cout)
{
    // Measured on the long-row launches of the bench step (profiles/r05_seq_gemm_sweep.txt; sweep over T, the regions' sizes):
    //   * T ~ half the number of rounds the tiles make over the chip's 512 workgroup slots, at least 2, at most the channel tiles, nudged
    //     to a divisor of the channel-tile count (a ragged last group is a short workgroup in the MIDDLE of the dispatch order);
    //   * from T = 4 on, ~512 tiles in sequences of 2 before the tail; ~512 single tiles as the tail once a point tile has >= 4 channel
    //     tiles (with 2-3 channel tiles per point tile the tail of single tiles measured slower than none).
    const int64_t n_pt = ceil_div(rows, 128), n_ct = ceil_div(cout, 128), tiles = n_pt * n_ct;
    if (n_ct < 2 || tiles < 1024) return 1;
    if (g_seq_lin && n_ct <= 8) {
        // round 6: balanced contiguous sequences (they may run from one point tile into the next).  Sequences of ~3 tiles, at least two
        // per workgroup slot: the dispatcher keeps something to hand out (one resident sequence per slot lost inside the
        // three-stream step in round 4), the slots drain together (lengths differ by one, the longer ones first)
        const int64_t per_slot = ceil_div(ceil_div(n_pt, 8) * n_ct, 64);       // tiles per slot of the fullest XCD
        const int64_t rounds = g_seq_lin == 2 ? 1 : std::min<int64_t>(15, std::max<int64_t>(2, (per_slot + 1) / 3));
        return (int)(0xF0 | rounds);
    }
    const int64_t cap = std::min<int64_t>(n_ct, 8);
    int64_t ta = std::min<int64_t>(cap, std::max<int64_t>(2, tiles / 1024));
    if (n_ct % ta != 0) {
        if (ta + 1 <= cap && n_ct % (ta + 1) == 0) ta += 1;
        else if (ta > 2 && n_ct % (ta - 1) == 0) ta -= 1;
    }
    const int64_t tb = ta >= 4 ? 2 : 1;
    const int64_t b_tiles = tb > 1 ? 512 : 0, c_tiles = n_ct >= 4 ? 512 : 0;
    return (int)(ta | tb << 4 | (b_tiles / 16) << 8 | (c_tiles / 16) << 16);
}

#define FFB6D_MLP_PM_ARGS                                                                                                    \
    bias, x1, k1, ld1, x1_idx, x1_rows_per_frame, x2, k2, ld2, y, ldy, y_idx, y_rows_per_frame, idx_bits, rows_per_frame, out, \
        ldo, rows, cout, act, tile_hint, stream

extern "C" int ffb6d_mlp_pm_f32(const float* w, const float* bias, const float* x1, int64_t k1, int64_t ld1, const void* x1_idx,
                                int64_t x1_rows_per_frame, const float* x2, int64_t k2, int64_t ld2, const float* y, int64_t ldy,
                                const void* y_idx, int64_t y_rows_per_frame, int idx_bits, int64_t rows_per_frame, float* out,
                                int64_t ldo, int64_t rows, int64_t cout, int act, int tile_hint, ffb6d_stream_t stream)
{
    return mlp_pm_impl<float>(w, FFB6D_MLP_PM_ARGS);
}

extern "C" int ffb6d_mlp_pm_bf16(const void* w, const float* bias, const void* x1, int64_t k1, int64_t ld1, const void* x1_idx,
                                 int64_t x1_rows_per_frame, const void* x2, int64_t k2, int64_t ld2, const void* y, int64_t ldy,
                                 const void* y_idx, int64_t y_rows_per_frame, int idx_bits, int64_t rows_per_frame, void* out,
                                 int64_t ldo, int64_t rows, int64_t cout, int act, int tile_hint, ffb6d_stream_t stream)
{
    return mlp_pm_impl<__bf16>(w, FFB6D_MLP_PM_ARGS);
}

extern "C" int ffb6d_att_pool_pm_f32(const float* w_fc, const float* f, int64_t c1, int64_t ldf, const void* nei, int idx_bits,
                                     const float* g, int64_t c2, int64_t ldg, float* out, int64_t ldo, int64_t B, int64_t N, int K,
                                     ffb6d_stream_t stream)
{
    return att_pool_pm_impl<float>(w_fc, f, c1, ldf, nei, idx_bits, g, c2, ldg, out, ldo, B, N, K, stream);
}

extern "C" int ffb6d_att_pool_pm_bf16(const void* w_fc, const void* f, int64_t c1, int64_t ldf, const void* nei, int idx_bits,
                                      const void* g, int64_t c2, int64_t ldg, void* out, int64_t ldo, int64_t B, int64_t N, int K,
                                      ffb6d_stream_t stream)
{
    return att_pool_pm_impl<__bf16>(w_fc, f, c1, ldf, nei, idx_bits, g, c2, ldg, out, ldo, B, N, K, stream);
}
