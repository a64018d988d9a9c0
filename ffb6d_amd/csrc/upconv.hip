// ffb6d_amd/csrc/upconv.hip -- the folded up-convolution of the colour branch's decoder for gfx950.
//
// Reference: PSPUpsample (ffb6d/models/cnn/pspnet.py:34-45): bilinear x2 (align_corners=True) -> Conv2d 3x3 -> BatchNorm ->
// PReLU.  The reference (and MIOpen behind it) convolves the UP-SAMPLED map: 725 + 181 + 181 GFLOP per batch of 8 frames,
// a third of all dense-convolution work of the network.  Up-sampling and channel mixing commute (csrc/upconv_body.h), so the
// channel mixing runs at the low resolution as ONE point-major GEMM with 9 * cout output channels (csrc/mlp_pm.hip: a
// quarter of the flops, on the hand-written MFMA kernel) and this file's kernel gathers the nine tap planes back together:
// HBM-bound, z is read once from HBM (every element is used by ~16 output pixels: L1/L2 hits), the result written once,
// with the BatchNorm shift and the PReLU in the same pass (the separate bilinear, BatchNorm and PReLU passes disappear).
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "ffb6d_ops.h"
#include "mfma_pm.h"
#include "upconv_body.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

// logical workgroup (bx, by) through the XCD-aware order (upconv::xcd_band_block): by = output row (b, Y) -- the vertical
// taps, source rows and weights are wave-uniform scalars -- and bx * 256 + thread = (output column, 16-byte unit) of that row
template <typename T>
__global__ void __launch_bounds__(BLK)
upconv_combine_pm_kernel(const upconv::CombineArgs a)
{
    unsigned bx, by;
    if (!upconv::xcd_band_block(blockIdx.x, a.nbx, a.nby, bx, by, a.banded != 0)) return;
    upconv::combine_body<T>(a, (int)by, (int)(bx * BLK + threadIdx.x));
}

// register-blocked form (exact x2 maps): by = (b, pair of output rows), bx * 256 + thread = (block of 4 output columns, unit)
template <typename T, int FORM>
__global__ void __launch_bounds__(BLK)
upconv_combine_block_pm_kernel(const upconv::CombineArgs a)
{
    unsigned bx, by;
    if (!upconv::xcd_band_block(blockIdx.x, a.nbx, a.nby, bx, by, a.banded != 0)) return;
    upconv::combine_block_body<T, FORM>(a, (int)by, (int)(bx * BLK + threadIdx.x));
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-staged form (round 6; exact x2 maps, both precisions).  The per-thread forms above are latency-bound: nine dependent rounds of
// window loads per thread at two or three waves per SIMD (3.0 / 1.3 TB/s of algorithmic bytes in fp32 / bf16,
// profiles/r06_upconv_probe_v1.txt -- the same time per ELEMENT in both precisions).  Here a workgroup owns 8 x 16 output pixels of ONE
// 64-byte channel chunk and
//   * stages the 7 x 11 low-resolution pixels x 9 taps its outputs blend (64 bytes each) into LDS with LDS-DMA loads -- no staging
//     registers, every request of the window in flight at once, one barrier;
//   * then every thread blends two (output pixel, 16-byte unit) items out of LDS: 36 ds_read_b128 per item, short latency, all in
//     flight together; three workgroups per CU (45 KB of LDS each) overlap one's staging with the others' blending.
// The arithmetic of an output element is combine_body's, operation for operation (same source elements, same order): equal bits.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LT_H = 8, LT_W = 16;                   // output tile
constexpr int LW_H = LT_H / 2 + 3, LW_W = LT_W / 2 + 3;      // source window (rows / columns): scale < 1/2, taps reach one position out
constexpr int L_CB = 64;                              // bytes of a channel chunk
constexpr int L_SEG = LW_H * LW_W * 9;                // 64-byte segments of a window: (pixel, tap)
constexpr int L_INSTR = (L_SEG + 63) / 64;            // LDS-DMA instructions per wave (4 waves x 16 segments each)
constexpr int L_LDS = L_INSTR * 64 * L_CB;

template <typename T>
__global__ void __launch_bounds__(BLK, 3)            // three workgroups per CU (LDS allows three): at most 168 registers per lane
upconv_combine_lds_kernel(const upconv::CombineArgs a, const int nty, const int nchunk)
{
    using U = upconv::Unit<T>;
    constexpr int SZ = 16 / U::VL;                    // bytes per element
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned bx, by;
    if (!upconv::xcd_band_block(blockIdx.x, a.nbx, a.nby, bx, by, true)) return;
    const int chunk = (int)(bx % (unsigned)nchunk), tx = (int)(bx / (unsigned)nchunk);
    const int ty = (int)(by % (unsigned)nty), b = (int)(by / (unsigned)nty);
    const int Y0 = ty * LT_H, X0 = tx * LT_W;
    const int hy0 = (int)(a.rh * (float)(Y0 > 0 ? Y0 - 1 : 0)), wx0 = (int)(a.rw * (float)(X0 > 0 ? X0 - 1 : 0));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pixb = 9 * a.q * 16;                    // bytes of a low-resolution pixel: 9 taps x C channels
    const __amdgpu_buffer_rsrc_t rs_z = pm::make_rsrc(a.z, (unsigned)((size_t)a.nby / nty * a.IH * a.IW * pixb));

    // stage: instruction k of wave w fills segments (4 k + w) * 16 .. + 15; lane -> segment + (lane >> 2), 16-byte piece lane & 3
#pragma unroll
    for (int k = 0; k < L_INSTR; ++k) {
        const int seg = (4 * k + wave) * 16 + (lane >> 2);
        const int pix = seg / 9, tap = seg - 9 * pix;
        const int wr = pix / LW_W, wc = pix - LW_W * wr;
        const int row = min(hy0 + wr, a.IH - 1), col = min(wx0 + wc, a.IW - 1);       // clamped positions are never addressed
        const int off = ((b * a.IH + row) * a.IW + col) * pixb + (tap * a.q * 16 + chunk * L_CB + (lane & 3) * 16);
        pm::lds_dma16(rs_z, lds + (4 * k + wave) * (16 * L_CB), seg < L_SEG ? off : 0x7ffffff0, 0);
    }
    __syncthreads();                                  // (vmcnt(0) rides in the barrier's fence: every wave's requests have landed)

    const float* shift = a.shift + (size_t)chunk * (L_CB / SZ);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int item = pass * BLK + threadIdx.x;    // (y, x, unit): unit fastest -- a pixel's 64 bytes are 4 consecutive lanes
        const int u = item & 3, x = (item >> 2) & (LT_W - 1), y = item >> 6;
        const int oy = Y0 + y, ox = X0 + x;
        if (oy >= a.OH || ox >= a.OW) continue;
        float acc[U::VL];
#pragma unroll
        for (int e = 0; e < U::VL; ++e) acc[e] = 0.f;
#pragma unroll 1                                      // (a filter row at a time: twelve reads in flight; unrolled, the bf16 form spills)
        for (int ky = 0; ky < 3; ++ky) {
            const int yr = oy + ky - 1;
            const bool y_in = yr >= 0 && yr < a.OH;
            const int yp = y_in ? yr : oy;
            const float h1r = a.rh * (float)yp;       // ATen upsample_bilinear2d, align_corners: as combine_body
            const int h1 = (int)h1r;
            const int h1p = (h1 < a.IH - 1) ? 1 : 0;
            const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
            const int r0 = (h1 - hy0) * LW_W, r1 = r0 + h1p * LW_W;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xr = ox + kx - 1;
                const bool in = y_in && xr >= 0 && xr < a.OW;
                const int xp = (xr >= 0 && xr < a.OW) ? xr : ox;
                const float w1r = a.rw * (float)xp;
                const int w1 = (int)w1r;
                const int w1p = (w1 < a.IW - 1) ? 1 : 0;
                const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
                const int c0 = w1 - wx0, c1 = c0 + w1p;
                const int t16 = (ky * 3 + kx) * L_CB + u * 16;
                auto at = [&](int pix) { return U::unpack(*reinterpret_cast<const uint4*>(lds + pix * (9 * L_CB) + t16)); };
                const U p00 = at(r0 + c0), p01 = at(r0 + c1), p10 = at(r1 + c0), p11 = at(r1 + c1);
#pragma unroll
                for (int e = 0; e < U::VL; ++e) {
                    const float v = upconv::blend2<T>(h0l, upconv::blend2<T>(w0l, p00.v[e], w1l, p01.v[e]), h1l, upconv::blend2<T>(w0l, p10.v[e], w1l, p11.v[e]));
                    acc[e] += in ? v : 0.f;
                }
            }
        }
        U o;
#pragma unroll
        for (int e = 0; e < U::VL; e += 4) {
            const float4 s4 = *reinterpret_cast<const float4*>(shift + u * U::VL + e);
            const float sh[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = acc[e + i] + sh[i];
                o.v[e + i] = v >= 0.f ? v : a.slope * v;        // PReLU
            }
        }
        o.store(a.out, ((size_t)(b * a.OH + oy) * a.OW + ox) * a.q + chunk * 4 + u);
    }
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

// A/B switch: 2 = the measured choice per shape (default); 3 = the LDS-staged form on every exact x2 map; 1 = the per-thread 2 x 4 block
// form (bf16 on half units); 0 = as 1 in fp32, one output pixel per thread in bf16 (rounds 2-5)
static int g_form = 2;
extern "C" void ffb6d_upconv_set_form(int form) { g_form = form; }

extern "C" int ffb6d_upconv_combine_pm(int dtype, const void* z, const float* shift, float slope, void* out, int64_t B, int64_t IH,
                                       int64_t IW, int64_t OH, int64_t OW, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dtype == 0 || dtype == 1, "upconv_combine_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype == 1 ? 8 : 4;
    FFB6D_REQUIRE(B >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1 && C >= VL && C % VL == 0,
                  "upconv_combine_pm: bad shape (C must be a positive multiple of %d)", VL);
    FFB6D_REQUIRE(IH < (1 << 24) && IW < (1 << 24) && OH < (1 << 24) && OW < (1 << 24), "upconv_combine_pm: too large");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(z && shift && out && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out) |
                                         reinterpret_cast<uintptr_t>(shift)) & 15) == 0,
                  "upconv_combine_pm: null or unaligned pointer");
    FFB6D_REQUIRE(ceil_div(OW * (C / VL), BLK) * B * OH < (1LL << 31) - 8, "upconv_combine_pm: too many workgroups");
    FFB6D_REQUIRE(OW * (C / VL) < (1LL << 31), "upconv_combine_pm: row too long");
    upconv::CombineArgs a;
    a.z = z; a.shift = shift; a.out = out;
    a.IH = (int)IH; a.IW = (int)IW; a.OH = (int)OH; a.OW = (int)OW;
    a.q = (int)(C / VL);
    a.rh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;       // ATen area_pixel_compute_scale, align_corners
    a.rw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    a.slope = slope;
    a.banded = 1;         // XCD-band workgroup order (measured +0-6 %, profiles/r03_upconv_blend_forms_ab.txt)
    const int64_t zbytes = B * IH * IW * 9 * C * (dtype == 1 ? 2 : 4);
    // Which form (profiles/r06_upconv_probe_v3_fma.txt, us per launch at bs = 8 fp32 / bs = 16 bf16: LDS-staged | 2 x 4 block | one pixel;
    // fp32 with the unfused blend, profiles/r06_upconv_probe_v2.txt):
    //   fp32 C = 256: 269 | 177;  fp32 C = 64: 263 | 194;  bf16 C = 256: 330 | 266 | 337;  bf16 C = 64: 275 | 294 | 346
    // -- every form is bound by vector-ALU issue (profiles/r06_upconv_pmc_*.txt: 93-160 lane operations per output element at 4 cycles
    // per wave instruction, 60 % of the SIMD cycles), the LDS-staged one wins where a pixel's channels are one chunk or two
    const bool lds_form = g_form == 3 || (g_form == 2 && dtype == 1 && C <= 64);
    if (lds_form && OH == 2 * IH && OW == 2 * IW && (C * (dtype == 1 ? 2 : 4)) % L_CB == 0 && zbytes < (1LL << 31) - 4096) {
        // LDS-staged form: 8 x 16 output pixels of one 64-byte channel chunk per workgroup
        const int nty = (int)ceil_div(OH, LT_H), ntx = (int)ceil_div(OW, LT_W), nchunk = (int)(C * (dtype == 1 ? 2 : 4) / L_CB);
        a.nbx = (unsigned)(ntx * nchunk);
        a.nby = (unsigned)(B * nty);
        FFB6D_REQUIRE((int64_t)a.nbx * a.nby < (1LL << 31) - 8, "upconv_combine_pm: too many workgroups");
        const dim3 grid((unsigned)(8 * ceil_div((int64_t)a.nbx * a.nby, 8)));
        if (dtype == 1)
            hipLaunchKernelGGL((upconv_combine_lds_kernel<__bf16>), grid, dim3(BLK), L_LDS, as_stream(stream), a, nty, nchunk);
        else
            hipLaunchKernelGGL((upconv_combine_lds_kernel<float>), grid, dim3(BLK), L_LDS, as_stream(stream), a, nty, nchunk);
    } else if (OH == 2 * IH && OW == 2 * IW && OW % 4 == 0 && (dtype == 0 || g_form >= 1)) {
        // bf16 rows: with 8-channel units the 2 x 4 block does not fit the register file without spilling -- the block form runs on
        // HALF units (4 channels in 8 bytes, RowUnit<Bf16Half>: the register footprint of the fp32 form)
        if (dtype == 1) a.q = (int)(C / 4);
        a.nbx = (unsigned)ceil_div(OW / 4 * (int64_t)a.q, BLK);
        a.nby = (unsigned)(B * OH / 2);
        const dim3 grid((unsigned)(8 * ceil_div((int64_t)a.nbx * a.nby, 8)));
        // 2 x 4 output pixels per thread with the compile-time operand pattern per tap: 2.2-2.8 TB/s against 1.5-1.8 for one pixel
        // per thread and 1.9-2.4 for per-element operand selects (profiles/r02_upconv_blend_forms_ab.txt; both A/B forms removed)
        if (dtype == 1)
            hipLaunchKernelGGL((upconv_combine_block_pm_kernel<Bf16Half, 2>), grid, dim3(BLK), 0, as_stream(stream), a);
        else
            hipLaunchKernelGGL((upconv_combine_block_pm_kernel<float, 2>), grid, dim3(BLK), 0, as_stream(stream), a);
    } else {
        a.nbx = (unsigned)ceil_div(OW * (int64_t)a.q, BLK);
        a.nby = (unsigned)(B * OH);
        const dim3 grid((unsigned)(8 * ceil_div((int64_t)a.nbx * a.nby, 8)));
        if (dtype == 1)
            hipLaunchKernelGGL((upconv_combine_pm_kernel<__bf16>), grid, dim3(BLK), 0, as_stream(stream), a);
        else
            hipLaunchKernelGGL((upconv_combine_pm_kernel<float>), grid, dim3(BLK), 0, as_stream(stream), a);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
