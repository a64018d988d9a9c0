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

#ifndef FFB6D_UPCONV_KX_UNROLL
#define FFB6D_UPCONV_KX_UNROLL _Pragma("unroll 1")
#endif

namespace ffb6d {
namespace upconv {

// a 16-byte unit of a row: VL consecutive channels, held as fp32 (the twin of ops_pm.hip's Unit, host-callable)
template <typename T> struct Unit;
template <> struct Unit<float> {
    static constexpr int VL = 4;
    float v[4];
    static __host__ __device__ __forceinline__ Unit load(const void* base, size_t unit)
    {
        const float4 f = static_cast<const float4*>(base)[unit];
        Unit u; u.v[0] = f.x; u.v[1] = f.y; u.v[2] = f.z; u.v[3] = f.w;
        return u;
    }
    __host__ __device__ __forceinline__ void store(void* base, size_t unit) const
    {
        static_cast<float4*>(base)[unit] = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Unit<__bf16> {
    static constexpr int VL = 8;
    float v[8];
    static __host__ __device__ __forceinline__ Unit load(const void* base, size_t unit)
    {
        const uint4 w = static_cast<const uint4*>(base)[unit];
        const unsigned int x[4] = {w.x, w.y, w.z, w.w};
        Unit u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u.v[2 * i] = __builtin_bit_cast(float, x[i] << 16);
            u.v[2 * i + 1] = __builtin_bit_cast(float, x[i] & 0xffff0000u);
        }
        return u;
    }
    __host__ __device__ __forceinline__ void store(void* base, size_t unit) const
    {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (__bf16)v[i];       // round to nearest even
        static_cast<bf16x8*>(base)[unit] = b;
    }
};

struct CombineArgs {
    const void* z;        // [B, IH, IW, 9, C] rows of T: tap-major blocks of C channels per low-resolution pixel
    const float* shift;   // [C] fp32: BatchNorm shift + BatchNorm scale * conv bias
    void* out;            // [B, OH, OW, C]
    int IH, IW, OH, OW;
    int q;                // 16-byte units per C channels
    float rh, rw;         // ATen's align_corners scales (IH-1)/(OH-1), (IW-1)/(OW-1)
    float slope;          // PReLU slope (one parameter)
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
                const float v = h0l * (w0l * p00.v[e] + w1l * p01.v[e]) + h1l * (w0l * p10.v[e] + w1l * p11.v[e]);
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

// Register-blocked form for the exact x2 case (OH = 2 IH, OW = 2 IW, OW % BX == 0): a thread owns BY x BX = 2 x 4 output
// pixels of one 16-byte channel unit.  combine_body issues 36 sixteen-byte loads per output unit and is bound by the L1
// (64 B/clk/CU: measured 1.9 TB/s of algorithmic bytes); here the BY x BX taps of one (ky, kx) share a window of NR x NC =
// 3 x 4 low-resolution pixels (scale < 1/2: two consecutive rows of the up-sampled map start at most one source row apart,
// four consecutive columns at most two source columns), i.e. 12 loads per 8 outputs and tap instead of 32.
// Every output is produced by the same operations in the same order as in combine_body -- horizontal blend of the two
// source rows, vertical blend, tap by tap -- with the operands picked out of the window by selects: bit-identical results
// (tests/test_hostsim_cpu.py), NaN / Inf stay confined to the pixels the reference spreads them to.
template <typename T, int BY, int BX>
__host__ __device__ __forceinline__ void combine_block_body(const CombineArgs& a, int rowblk, int t)
{
    using U = Unit<T>;
    constexpr int NR = BY / 2 + 2, NC = BX / 2 + 2;
    const int XB = a.OW / BX, RB = a.OH / BY;
    if (t >= XB * a.q) return;
    const int yb = rowblk % RB, b = rowblk / RB;
    const int xb = t / a.q;
    const int c = t - xb * a.q;
    const int Y0 = yb * BY, X0 = xb * BX;
    const size_t q9 = (size_t)9 * a.q;
    float acc[BY][BX][U::VL];
#pragma unroll
    for (int i = 0; i < BY; ++i)
#pragma unroll
        for (int j = 0; j < BX; ++j)
#pragma unroll
            for (int e = 0; e < U::VL; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        // the BY rows of the up-sampled map this tap reads, their source rows and weights (as combine_body)
        bool y_in[BY];
        int ra[BY], rb[BY];                         // window rows of the two source rows of output row i
        float h0l[BY], h1l[BY];
        int hb = 0;
#pragma unroll
        for (int i = 0; i < BY; ++i) {
            const int yr = Y0 + i + ky - 1;
            y_in[i] = yr >= 0 && yr < a.OH;
            const int yp = y_in[i] ? yr : Y0 + i;
            const float h1r = a.rh * (float)yp;
            const int h1 = (int)h1r;
            const int h1p = (h1 < a.IH - 1) ? 1 : 0;
            h1l[i] = h1r - (float)h1;
            h0l[i] = 1.f - h1l[i];
            if (i == 0) hb = h1;
            ra[i] = h1 - hb;
            rb[i] = ra[i] + h1p;
        }
FFB6D_UPCONV_KX_UNROLL
        for (int kx = 0; kx < 3; ++kx) {
            bool x_in[BX];
            int ca[BX], cb[BX];
            float w0l[BX], w1l[BX];
            int wb = 0;
#pragma unroll
            for (int j = 0; j < BX; ++j) {
                const int xr = X0 + j + kx - 1;
                x_in[j] = xr >= 0 && xr < a.OW;
                const int xp = x_in[j] ? xr : X0 + j;
                const float w1r = a.rw * (float)xp;
                const int w1 = (int)w1r;
                const int w1p = (w1 < a.IW - 1) ? 1 : 0;
                w1l[j] = w1r - (float)w1;
                w0l[j] = 1.f - w1l[j];
                if (j == 0) wb = w1;
                ca[j] = w1 - wb;
                cb[j] = ca[j] + w1p;
            }
            // the window of this tap: NR x NC low-resolution pixels from (hb, wb), clamped to the map (a clamped
            // element is never selected: the second source row / column coincides with the first one at the border)
            U win[NR][NC];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int hr = hb + r < a.IH ? hb + r : a.IH - 1;
                const size_t rowoff = ((size_t)b * a.IH + hr) * a.IW * q9 + (size_t)(ky * 3 + kx) * a.q + c;
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    const int wc = wb + cc < a.IW ? wb + cc : a.IW - 1;
                    win[r][cc] = U::load(a.z, rowoff + (size_t)wc * q9);
                }
            }
            // horizontal blend of every window row for the BX output columns
            float s[NR][BX][U::VL];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int j = 0; j < BX; ++j)
#pragma unroll
                    for (int e = 0; e < U::VL; ++e) {
                        float pa = win[r][0].v[e], pb = win[r][0].v[e];
#pragma unroll
                        for (int cc = 1; cc < NC; ++cc) {
                            pa = ca[j] == cc ? win[r][cc].v[e] : pa;
                            pb = cb[j] == cc ? win[r][cc].v[e] : pb;
                        }
                        s[r][j][e] = w0l[j] * pa + w1l[j] * pb;
                    }
#pragma unroll
            for (int i = 0; i < BY; ++i)
#pragma unroll
                for (int j = 0; j < BX; ++j) {
                    const bool in = y_in[i] && x_in[j];
#pragma unroll
                    for (int e = 0; e < U::VL; ++e) {
                        float sa = s[0][j][e], sb = s[0][j][e];
#pragma unroll
                        for (int r = 1; r < NR; ++r) {
                            sa = ra[i] == r ? s[r][j][e] : sa;
                            sb = rb[i] == r ? s[r][j][e] : sb;
                        }
                        const float v = h0l[i] * sa + h1l[i] * sb;
                        acc[i][j][e] += in ? v : 0.f;
                    }
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
    for (int i = 0; i < BY; ++i)
#pragma unroll
        for (int j = 0; j < BX; ++j) {
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
