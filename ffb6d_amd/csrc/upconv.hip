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

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

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
    if (OH == 2 * IH && OW == 2 * IW && OW % 4 == 0 && dtype == 0) {
        // fp32 rows only: with 8-channel bf16 units the 2 x 4 block does not fit the register file without spilling
        a.nbx = (unsigned)ceil_div(OW / 4 * (int64_t)a.q, BLK);
        a.nby = (unsigned)(B * OH / 2);
        const dim3 grid((unsigned)(8 * ceil_div((int64_t)a.nbx * a.nby, 8)));
        // 2 x 4 output pixels per thread with the compile-time operand pattern per tap: 2.2-2.8 TB/s against 1.5-1.8 for one pixel
        // per thread and 1.9-2.4 for per-element operand selects (profiles/r02_upconv_blend_forms_ab.txt; both A/B forms removed)
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
