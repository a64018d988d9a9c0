// ffb6d_amd/csrc/upconv_body.h -- per-thread bodies of the folded up-convolution (csrc/upconv.hip).
//
// Reference: PSPUpsample = Upsample(x2, bilinear, align_corners=True) -> Conv2d(cin, cout, 3, padding=1) -> BatchNorm2d
// -> PReLU (ffb6d/models/cnn/pspnet.py:34-45), applied at 60x80 -> 120x160 (1024 -> 256), 120x160 -> 240x320 (256 -> 64)
// and 240x320 -> 480x640 (64 -> 64) (pspnet.py:57-59, ffb6d.py:86-87).
//
// Both the up-sampling U (a per-channel spatial operator) and the channel mixing W_tap of every filter tap are linear
// and commute, so with z_tap = (BN scale * W_tap) x computed at the LOW resolution (one GEMM, 9 * cout output channels,
// a quarter of the pixels: 4x fewer flops than the convolution of the up-sampled map)
//
//     out(Y, X, :) = prelu( shift + sum_{ky,kx} [ (Y+ky-1, X+kx-1) inside the up-sampled map ] * (U z_tap)(Y+ky-1, X+kx-1, :) )
//
// The bodies are plain per-thread code without cross-lane operations or LDS, written __host__ __device__ so that the
// CPU test-suite can run the very same source on the host (tests/hostsim/) against torch's upsample + conv2d.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "row_unit.h"

namespace ffb6d {
namespace upconv {

template <typename T> using Unit = RowUnit<T>;      // csrc/row_unit.h

// a0 * x0 + a1 * x1.  Every form of the blend is bound by vector-ALU issue (profiles/r06_upconv_pmc_*.txt), so the instruction count IS the
// time -- and it depends on the element type (profiles/r06_upconv_probe_v3_fma.txt): on bfloat16 units one product + one fused
// multiply-add is a quarter fewer instructions (2 x 4 block 363 -> 266 us, 418 -> 294 us); on float32 units hipcc packs the unfused
// products and sums of neighbouring channels into v_pk_mul_f32 / v_pk_add_f32 (1.5 instructions per element) and does not pack the
// fused form (2 per element): 177 -> 195 us.  So: fused for bfloat16 (and its half units), unfused for float32.  Every body below blends
// through this function (horizontal pair, then vertical pair, tap by tap): the forms of one precision stay bit-identical to one another.
template <typename T>
__host__ __device__ __forceinline__ float blend2(float a0, float x0, float a1, float x1)
{
    if constexpr (sizeof(T) == sizeof(float)) return a0 * x0 + a1 * x1;
    else return fmaf(a1, x1, a0 * x0);
}

// XCD-aware order of a 2-D launch (MI355X: 8 XCDs with private L2s, workgroup i of a 1-D grid runs on XCD i % 8 -- observed
// dispatch rule, speed only): XCD x gets the x-th contiguous eighth of the row-major (by, bx) blocks, i.e. a band of
// consecutive output rows, so the source rows a band re-reads stay in ONE L2 instead of being fetched by all eight.
// `id` = index in a grid of 8 * ceil(nbx * nby / 8) workgroups; false = surplus workgroup.
__host__ __device__ __forceinline__ bool xcd_band_block(unsigned id, unsigned nbx, unsigned nby, unsigned& bx, unsigned& by,
                                                        bool banded = true)
{
    const unsigned nb = nbx * nby, per = (nb + 7u) >> 3;
    if (!banded) {                                   // plain row-major order (A/B)
        by = id / nbx;
        bx = id - by * nbx;
        return id < nb;
    }
    const unsigned l = (id & 7u) * per + (id >> 3);
    if ((id >> 3) >= per || l >= nb) return false;
    by = l / nbx;
    bx = l - by * nbx;
    return true;
}

struct CombineArgs {
    const void* z;        // [B, IH, IW, 9, C] rows of T: tap-major blocks of C channels per low-resolution pixel
    const float* shift;   // [C] fp32: BatchNorm shift + BatchNorm scale * conv bias
    void* out;            // [B, OH, OW, C]
    int IH, IW, OH, OW;
    int q;                // 16-byte units per C channels
    float rh, rw;         // ATen's align_corners scales (IH-1)/(OH-1), (IW-1)/(OW-1)
    float slope;          // PReLU slope (one parameter)
    int banded;           // 1: XCD-band workgroup order (default), 0: row-major (the A/B form of round 3)
    unsigned nbx, nby;    // logical launch: workgroups along the row of threads, output rows (or row pairs) of all frames
};

// row = b * OH + Y (uniform over a workgroup), t = X * q + unit
template <typename T>
__host__ __device__ __forceinline__ void combine_body(const CombineArgs& a, int row, int t)
{
    using U = Unit<T>;
    if (t >= a.OW * a.q) return;
    const int oy = row % a.OH, b = row / a.OH;
    const int ox = t / a.q;
    const int c = t - ox * a.q;
    const size_t q9 = (size_t)9 * a.q;             // units per low-resolution pixel
    float acc[U::VL];
#pragma unroll
    for (int e = 0; e < U::VL; ++e) acc[e] = 0.f;
    // Branch-free: taps that fall into the zero padding of the up-sampled map read a clamped (valid) address and are
    // dropped by a select, so that the loads of all nine taps can be in flight together.
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yr = oy + ky - 1;                 // row of the up-sampled map this tap reads (zero padding outside)
        const bool y_in = yr >= 0 && yr < a.OH;
        const int yp = y_in ? yr : oy;
        // ATen upsample_bilinear2d, align_corners: source index = scale * dst; as bilinear_pm_kernel (csrc/ops_pm.hip)
        const float h1r = a.rh * (float)yp;
        const int h1 = (int)h1r;
        const int h1p = (h1 < a.IH - 1) ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
        const size_t r0 = ((size_t)b * a.IH + h1) * a.IW * q9;
        const size_t r1 = r0 + (size_t)h1p * a.IW * q9;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xr = ox + kx - 1;
            const bool in = y_in && xr >= 0 && xr < a.OW;
            const int xp = (xr >= 0 && xr < a.OW) ? xr : ox;
            const float w1r = a.rw * (float)xp;
            const int w1 = (int)w1r;
            const int w1p = (w1 < a.IW - 1) ? 1 : 0;
            const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
            const size_t i0 = (size_t)w1 * q9 + (size_t)(ky * 3 + kx) * a.q + c, i1 = i0 + (size_t)w1p * q9;
            const U p00 = U::load(a.z, r0 + i0), p01 = U::load(a.z, r0 + i1), p10 = U::load(a.z, r1 + i0), p11 = U::load(a.z, r1 + i1);
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                const float v = blend2<T>(h0l, blend2<T>(w0l, p00.v[e], w1l, p01.v[e]), h1l, blend2<T>(w0l, p10.v[e], w1l, p11.v[e]));
                acc[e] += in ? v : 0.f;
            }
        }
    }
    U o;
#pragma unroll
    for (int e = 0; e < U::VL; e += 4) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.shift + (size_t)c * U::VL + e);
        const float s[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = acc[e + i] + s[i];
            o.v[e + i] = v >= 0.f ? v : a.slope * v;        // PReLU
        }
    }
    o.store(a.out, (size_t)row * a.OW * a.q + t);
}

// Register-blocked form for the exact x2 case (OH = 2 IH, OW = 2 IW, OW % 4 == 0): a thread owns 2 x 4 output pixels of one
// 16-byte channel unit.  combine_body issues 36 sixteen-byte loads per output unit and is bound by the L1 (64 B/clk/CU:
// measured 1.9 TB/s of algorithmic bytes); here the 2 x 4 positions one filter tap reads share a window of at most 3 x 4
// low-resolution pixels (scale < 1/2: two consecutive rows of the up-sampled map start at most one source row apart, four
// consecutive columns at most two source columns).
// Every output is produced by the same operations in the same order as in combine_body -- horizontal blend of the two
// source rows, vertical blend, tap by tap -- so the results are bit-identical (tests/test_hostsim_cpu.py) and NaN / Inf
// stay confined to the pixels the reference spreads them to.  Two ways to pick the operands out of the window:
//   * tap_select: per-element selects on the window position of each operand (any block; ~900 vector instructions per tap);
//   * tap_static: in exact arithmetic position 2m of the up-sampled axis reads source pixels (m-1, m) and 2m+1 reads
//     (m, m+1), so for an aligned block the window positions are compile-time constants (and the window shrinks to 2..3 rows
//     x 3..4 columns: 70 loads per thread instead of 108).  The float arithmetic of ATen can deviate from that pattern only
//     where a source index lands on an integer (the last row / column), and the first block of an axis is clamped: a thread
//     takes the static path for a tap only after checking that the positions it computed ARE the pattern, else tap_select.
struct Axis2 { bool in[2]; int a[2], b[2]; float l0[2], l1[2]; int base; };
struct Axis4 { bool in[4]; int a[4], b[4]; float l0[4], l1[4]; int base; };

// positions P0 + n + k - 1 (n < B) of an up-sampled axis of length O over a source axis of length I: inside the map?, window
// offsets of the two source pixels, blend weights (ATen upsample_bilinear2d, align_corners: as combine_body)
template <int B, typename AX>
__host__ __device__ __forceinline__ void tap_axis(AX& ax, int P0, int k, int O, int I, float scale)
{
#pragma unroll
    for (int n = 0; n < B; ++n) {
        const int pr = P0 + n + k - 1;
        ax.in[n] = pr >= 0 && pr < O;
        const int pp = ax.in[n] ? pr : P0 + n;
        const float sr = scale * (float)pp;
        const int s1 = (int)sr;
        const int s1p = (s1 < I - 1) ? 1 : 0;
        ax.l1[n] = sr - (float)s1;
        ax.l0[n] = 1.f - ax.l1[n];
        if (n == 0) ax.base = s1;
        ax.a[n] = s1 - ax.base;
        ax.b[n] = ax.a[n] + s1p;
    }
}

template <typename T>
__host__ __device__ __forceinline__ void tap_select(const CombineArgs& a, const Axis2& ay, const Axis4& ax, size_t tapoff,
                                                    size_t q9, int b, float (&acc)[2][4][Unit<T>::VL])
{
    using U = Unit<T>;
    constexpr int NR = 3, NC = 4;
    // the window of this tap: NR x NC low-resolution pixels from (base row, base column), clamped to the map (a clamped
    // element is never selected: the second source row / column coincides with the first one at the border)
    U win[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int hr = ay.base + r < a.IH ? ay.base + r : a.IH - 1;
        const size_t rowoff = ((size_t)b * a.IH + hr) * a.IW * q9 + tapoff;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int wc = ax.base + cc < a.IW ? ax.base + cc : a.IW - 1;
            win[r][cc] = U::load(a.z, rowoff + (size_t)wc * q9);
        }
    }
    float s[NR][4][U::VL];                           // horizontal blend of every window row for the 4 output columns
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                float pa = win[r][0].v[e], pb = win[r][0].v[e];
#pragma unroll
                for (int cc = 1; cc < NC; ++cc) {
                    pa = ax.a[j] == cc ? win[r][cc].v[e] : pa;
                    pb = ax.b[j] == cc ? win[r][cc].v[e] : pb;
                }
                s[r][j][e] = blend2<T>(ax.l0[j], pa, ax.l1[j], pb);
            }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = ay.in[i] && ax.in[j];
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                float sa = s[0][j][e], sb = s[0][j][e];
#pragma unroll
                for (int r = 1; r < NR; ++r) {
                    sa = ay.a[i] == r ? s[r][j][e] : sa;
                    sb = ay.b[i] == r ? s[r][j][e] : sb;
                }
                const float v = blend2<T>(ay.l0[i], sa, ay.l1[i], sb);
                acc[i][j][e] += in ? v : 0.f;
            }
        }
}

// the pattern of an aligned block: tap k = 1 starts on an even position (offsets 0,1,1,2..), taps 0 and 2 on an odd one (0,0,1,1..)
__host__ __device__ constexpr int pattern_a(bool mid, int n) { return mid ? (n + 1) >> 1 : n >> 1; }

// window of one tap in the static pattern: NR x NC source pixels from (base row, base column), every one of them a source
// pixel of some output, hence inside the map
template <typename T, int NR, int NC>
__host__ __device__ __forceinline__ void win_load(const CombineArgs& a, const Axis2& ay, const Axis4& ax, size_t tapoff, size_t q9,
                                                  int b, Unit<T> (&win)[NR][NC])
{
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t rowoff = ((size_t)b * a.IH + ay.base + r) * a.IW * q9 + tapoff;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) win[r][cc] = Unit<T>::load(a.z, rowoff + (size_t)(ax.base + cc) * q9);
    }
}

template <typename T, bool KYM, bool KXM, int NR, int NC>
__host__ __device__ __forceinline__ void win_blend(const Axis2& ay, const Axis4& ax, const Unit<T> (&win)[NR][NC],
                                                   float (&acc)[2][4][Unit<T>::VL])
{
    using U = Unit<T>;
    float s[NR][4][U::VL];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < U::VL; ++e)
                s[r][j][e] = blend2<T>(ax.l0[j], win[r][pattern_a(KXM, j)].v[e], ax.l1[j], win[r][pattern_a(KXM, j) + 1].v[e]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = ay.in[i] && ax.in[j];
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                const float v = blend2<T>(ay.l0[i], s[pattern_a(KYM, i)][j][e], ay.l1[i], s[pattern_a(KYM, i) + 1][j][e]);
                acc[i][j][e] += in ? v : 0.f;
            }
        }
}

template <typename T, bool KYM, bool KXM>
__host__ __device__ __forceinline__ void tap_static(const CombineArgs& a, const Axis2& ay, const Axis4& ax, size_t tapoff,
                                                    size_t q9, int b, float (&acc)[2][4][Unit<T>::VL])
{
    constexpr int NR = KYM ? 3 : 2, NC = KXM ? 4 : 3;
    Unit<T> win[NR][NC];
    win_load<T, NR, NC>(a, ay, ax, tapoff, q9, b, win);
    win_blend<T, KYM, KXM, NR, NC>(ay, ax, win, acc);
}

// rowblk = b * (OH / 2) + pair of output rows, t = (block of 4 output columns) * q + unit
// FORM: 1 = operand selects for every tap, 2 = compile-time pattern per tap where it holds.  (A third form that issued the three
// taps of a filter row together was measured and removed: 346 registers, one wave per SIMD, 1.5 - 1.8 TB/s against 2.2 - 2.8,
// profiles/r03_upconv_blend_forms_ab.txt.)
template <typename T, int FORM>
__host__ __device__ __forceinline__ void combine_block_body(const CombineArgs& a, int rowblk, int t)
{
    using U = Unit<T>;
    const int XB = a.OW / 4, RB = a.OH / 2;
    if (t >= XB * a.q) return;
    const int yb = rowblk % RB, b = rowblk / RB;
    const int xb = t / a.q;
    const int c = t - xb * a.q;
    const int Y0 = yb * 2, X0 = xb * 4;
    const size_t q9 = (size_t)9 * a.q;
    float acc[2][4][U::VL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < U::VL; ++e) acc[i][j][e] = 0.f;
    // the tap loops stay rolled: unrolled, the compiler keeps the loads of all nine windows in flight and spills
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        Axis2 ay;
        tap_axis<2>(ay, Y0, ky, a.OH, a.IH, a.rh);
        constexpr bool STATIC = FORM >= 2;
        bool rows_ok = STATIC;
#pragma unroll
        for (int i = 0; i < 2; ++i) rows_ok = rows_ok && ay.a[i] == pattern_a(ky == 1, i) && ay.b[i] == ay.a[i] + 1;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            Axis4 ax;
            tap_axis<4>(ax, X0, kx, a.OW, a.IW, a.rw);
            bool ok = rows_ok;
#pragma unroll
            for (int j = 0; j < 4; ++j) ok = ok && ax.a[j] == pattern_a(kx == 1, j) && ax.b[j] == ax.a[j] + 1;
            const size_t tapoff = (size_t)(ky * 3 + kx) * a.q + c;
            if (STATIC && ok) {
                if (ky == 1) {
                    if (kx == 1) tap_static<T, true, true>(a, ay, ax, tapoff, q9, b, acc);
                    else tap_static<T, true, false>(a, ay, ax, tapoff, q9, b, acc);
                } else {
                    if (kx == 1) tap_static<T, false, true>(a, ay, ax, tapoff, q9, b, acc);
                    else tap_static<T, false, false>(a, ay, ax, tapoff, q9, b, acc);
                }
            } else {
                tap_select<T>(a, ay, ax, tapoff, q9, b, acc);
            }
        }
    }
    float sh[U::VL];
#pragma unroll
    for (int e = 0; e < U::VL; e += 4) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.shift + (size_t)c * U::VL + e);
        sh[e] = s4.x; sh[e + 1] = s4.y; sh[e + 2] = s4.z; sh[e + 3] = s4.w;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            U o;
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                const float v = acc[i][j][e] + sh[e];
                o.v[e] = v >= 0.f ? v : a.slope * v;        // PReLU
            }
            o.store(a.out, ((size_t)(b * a.OH + Y0 + i) * a.OW + X0 + j) * a.q + c);
        }
}

}  // namespace upconv
}  // namespace ffb6d
