// ffb6d_amd/csrc/train_ops.hip -- backward bodies that dominate the training step once the convolutions run in bf16
// (profiles/r03_rocprofv3_kernel_stats_train_*.txt), for gfx950.  The reference trains FFB6D with stock autograd
// (train_lm.py:592-628); these replace the three slowest non-convolution backward passes of the colour decoder and of the
// point -> pixel fusion:
//
//   bilinear_bwd_pm   gradient of the align_corners bilinear up-sampling of PSPUpsample (pspnet.py:37-42) as a GATHER: a thread
//                     owns one 16-byte unit of an INPUT pixel and adds up the <= 3 x 3 output pixels that interpolate from it,
//                     with ATen's own source-index arithmetic (same weights as the forward, csrc/ops_pm.hip bilinear_pm_kernel).
//                     ATen's backward is an atomic scatter over 629 MB per block: 9.9 ms per step.
//   prelu_fwd / bwd   single-slope PReLU of PSPUpsample (pspnet.py:43): the slope's gradient is reduced inside the kernel (wave
//                     shuffle -> one atomic per wave) instead of materialising a gradient tensor as large as the map (ATen: 10 ms).
//   nearest_interpolation_bwd (csrc/neighbour_ops.hip keeps the entry point): scatter-add of the pixel gradients onto the few
//                     points they were interpolated from, privatised in LDS -- a workgroup owns (frame, channel group), keeps the
//                     gradient rows of its channels in LDS, adds with LDS atomics and writes the rows out once (the global-atomic
//                     form: 11.3 ms per step).
//
// Layout: bilinear / PReLU work on pixel-major rows [B,H,W,C] = torch's channels_last memory format of a [B,C,H,W] tensor
// (what bench.py --mode train keeps the colour branch in); fp32 or bf16 rows, fp32 arithmetic.
#include <algorithm>

#include "common.h"
#include "ffb6d_ops.h"
#include "row_unit.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

__device__ __forceinline__ float src_index_ac(float scale, int dst) { return scale * (float)dst; }      // align_corners = True

// weight with which output coordinate o (of an axis of length O, source length I, scale r) reads input coordinate i
__device__ __forceinline__ float axis_weight(float r, int o, int i, int I)
{
    const float s = src_index_ac(r, o);
    const int i0 = (int)s;
    const int i1 = i0 + ((i0 < I - 1) ? 1 : 0);
    const float l1 = s - (float)i0, l0 = 1.f - l1;
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// gin[b, iy, ix, :] = sum over (oy, ox) of wy(oy, iy) * wx(ox, ix) * gout[b, oy, ox, :]
// blockIdx.y = input row (b, iy); blockIdx.x * 256 + thread = (input column, unit).  Candidate output rows / columns: the
// window [lo, hi] that can reference the input coordinate (at most hi - lo + 1 <= WIN, host-checked); zero weights skip the load.
template <typename T, int WIN>
__global__ void __launch_bounds__(BLK)
bilinear_bwd_pm_kernel(const void* __restrict__ gout, void* __restrict__ gin, int IH, int IW, int OH, int OW, int q, float rh, float rw,
                       float inv_rh, float inv_rw)
{
    using U = RowUnit<T>;
    const int row = blockIdx.y;                  // b * IH + iy
    const int iy = row % IH, b = row / IH;
    const int t = blockIdx.x * BLK + threadIdx.x;
    if (t >= IW * q) return;
    const int ix = t / q, c = t - ix * q;
    // outputs that may read input coordinate i lie in ((i - 1) / r, (i + 1) / r): widen by one on both sides, test exactly
    const int oy_lo = max(0, (int)floorf((float)(iy - 1) * inv_rh) - 1), oy_hi = min(OH - 1, (int)ceilf((float)(iy + 1) * inv_rh) + 1);
    const int ox_lo = max(0, (int)floorf((float)(ix - 1) * inv_rw) - 1), ox_hi = min(OW - 1, (int)ceilf((float)(ix + 1) * inv_rw) + 1);
    float acc[U::VL];
#pragma unroll
    for (int e = 0; e < U::VL; ++e) acc[e] = 0.f;
    float wx[WIN];
#pragma unroll
    for (int j = 0; j < WIN; ++j) wx[j] = (ox_lo + j <= ox_hi) ? axis_weight(rw, ox_lo + j, ix, IW) : 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float wy = axis_weight(rh, oy, iy, IH);
        if (wy == 0.f) continue;                                     // row-uniform (blockIdx.y)
        const size_t base = ((size_t)b * OH + oy) * OW * q + c;
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
            if (wx[j] == 0.f) continue;
            const U g = U::load(gout, base + (size_t)(ox_lo + j) * q);
            const float w = wy * wx[j];
#pragma unroll
            for (int e = 0; e < U::VL; ++e) acc[e] = fmaf(w, g.v[e], acc[e]);
        }
    }
    U o;
#pragma unroll
    for (int e = 0; e < U::VL; ++e) o.v[e] = acc[e];
    o.store(gin, (size_t)row * IW * q + t);
}

// y = x > 0 ? x : a * x on n units
template <typename T>
__global__ void __launch_bounds__(BLK)
prelu_fwd_kernel(const void* __restrict__ x, const float* __restrict__ slope, void* __restrict__ y, size_t n_units)
{
    using U = RowUnit<T>;
    const float a = *slope;
    for (size_t t = (size_t)blockIdx.x * BLK + threadIdx.x; t < n_units; t += (size_t)gridDim.x * BLK) {
        U v = U::load(x, t);
#pragma unroll
        for (int e = 0; e < U::VL; ++e) v.v[e] = v.v[e] > 0.f ? v.v[e] : a * v.v[e];
        v.store(y, t);
    }
}

// gx = g * (x > 0 ? 1 : a);  *ga += sum over x <= 0 of g * x   (fp32 accumulation, one atomic per wave)
template <typename T>
__global__ void __launch_bounds__(BLK)
prelu_bwd_kernel(const void* __restrict__ x, const void* __restrict__ g, const float* __restrict__ slope, void* __restrict__ gx,
                 float* __restrict__ ga, size_t n_units)
{
    using U = RowUnit<T>;
    const float a = *slope;
    float part = 0.f;
    for (size_t t = (size_t)blockIdx.x * BLK + threadIdx.x; t < n_units; t += (size_t)gridDim.x * BLK) {
        const U xv = U::load(x, t);
        U gv = U::load(g, t);
#pragma unroll
        for (int e = 0; e < U::VL; ++e) {
            const bool pos = xv.v[e] > 0.f;
            part += pos ? 0.f : gv.v[e] * xv.v[e];
            gv.v[e] = pos ? gv.v[e] : a * gv.v[e];
        }
        gv.store(gx, t);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(ga, part);
}

// grad_feat[b, c, idx[b, u]] += grad_out[b, c, u] for the CG channels c0 .. c0 + CG - 1 of frame b, privatised in LDS:
// hist[CG][M] floats; thread walks u, adds its CG gradients with LDS atomics; rows leave once, coalesced (no memset needed).
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
nearest_interp_bwd_lds_kernel(const float* __restrict__ grad_out, const IdxT* __restrict__ idx, float* __restrict__ grad_feat, int C, int M,
                              int U, int CG)
{
    extern __shared__ __attribute__((aligned(16))) float hist[];            // [CG][M]
    const int b = blockIdx.y, c0 = blockIdx.x * CG;
    const int cg = min(CG, C - c0);
    for (int i = threadIdx.x; i < cg * M; i += BLK) hist[i] = 0.f;
    __syncthreads();
    const IdxT* ib = idx + (size_t)b * U;
    const float* gb = grad_out + ((size_t)b * C + c0) * U;
    for (int u = threadIdx.x; u < U; u += BLK) {
        const int i = (int)ib[u];
        for (int c = 0; c < cg; ++c) atomicAdd(&hist[c * M + i], gb[(size_t)c * U + u]);
    }
    __syncthreads();
    float* out = grad_feat + ((size_t)b * C + c0) * M;
    for (int i = threadIdx.x; i < cg * M; i += BLK) out[i] = hist[i];
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// called by ffb6d_nearest_interpolation_bwd_f32 (csrc/neighbour_ops.hip) when the privatised form fits: returns false otherwise
bool nearest_interp_bwd_lds(const float* grad_out, const void* idx, int idx_bits, float* grad_feat, int64_t B, int64_t C, int64_t M,
                            int64_t U, hipStream_t st)
{
    const int64_t budget = 96 << 10;                                   // bytes of LDS per workgroup
    if (M * 4 > budget || B >= 65536 || U >= (1LL << 31)) return false;
    int64_t CG = std::min<int64_t>(C, budget / (4 * M));
    // enough workgroups to fill the chip when there are channels to spare
    while (CG > 1 && B * ceil_div(C, CG) < 512) CG = (CG + 1) / 2;
    const size_t lds = (size_t)CG * M * 4;
    const dim3 grid((unsigned)ceil_div(C, CG), (unsigned)B);
    if (idx_bits == 64) {
        static int attr_set[kMaxDevices + 1];                                  // per device (common.h: device_slot)
        const int slot = device_slot();
        if (!cache_get(attr_set, slot)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nearest_interp_bwd_lds_kernel<int64_t>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10) != hipSuccess)
                return false;                                                  // the caller falls back to the atomic form
            cache_set(attr_set, slot, 1);
        }
        hipLaunchKernelGGL((nearest_interp_bwd_lds_kernel<int64_t>), grid, dim3(BLK), lds, st, grad_out, static_cast<const int64_t*>(idx),
                           grad_feat, (int)C, (int)M, (int)U, (int)CG);
    } else {
        static int attr_set[kMaxDevices + 1];                                  // per device (common.h: device_slot)
        const int slot = device_slot();
        if (!cache_get(attr_set, slot)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nearest_interp_bwd_lds_kernel<int32_t>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10) != hipSuccess)
                return false;                                                  // the caller falls back to the atomic form
            cache_set(attr_set, slot, 1);
        }
        hipLaunchKernelGGL((nearest_interp_bwd_lds_kernel<int32_t>), grid, dim3(BLK), lds, st, grad_out, static_cast<const int32_t*>(idx),
                           grad_feat, (int)C, (int)M, (int)U, (int)CG);
    }
    return true;
}

}  // namespace ffb6d

using namespace ffb6d;

#define FFB6D_TRAIN_DT(dtype, T, ...)            \
    do {                                         \
        if (dtype == 1) { using T = __bf16; __VA_ARGS__ } else { using T = float; __VA_ARGS__ } \
    } while (0)

extern "C" int ffb6d_bilinear_bwd_pm(int dtype, const void* grad_out, void* grad_in, int64_t B, int64_t IH, int64_t IW, int64_t OH,
                                     int64_t OW, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dtype == 0 || dtype == 1, "bilinear_bwd_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(B >= 0 && IH >= 1 && IW >= 1 && OH >= IH && OW >= IW && C >= VL && C % VL == 0,
                  "bilinear_bwd_pm: up-sampling only (OH >= IH, OW >= IW), C a multiple of %d", VL);
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(grad_out && grad_in && al16(grad_out) && al16(grad_in), "bilinear_bwd_pm: null or unaligned pointer");
    FFB6D_REQUIRE(B * IH < 65536 && OH < (1 << 24) && OW < (1 << 24), "bilinear_bwd_pm: too large");
    const float rh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f, rw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    // rows / columns of the output that can read one input coordinate: 2 / r + 3 candidates; the kernel's window holds 12
    FFB6D_REQUIRE((IH == 1 || rh >= 0.25f) && (IW == 1 || rw >= 0.25f) && IH > 1 && IW > 1,
                  "bilinear_bwd_pm: scale factors up to 4 and maps of at least 2 x 2 pixels");
    const int q = (int)(C / VL);
    const dim3 grid((unsigned)ceil_div(IW * (int64_t)q, BLK), (unsigned)(B * IH));
    FFB6D_TRAIN_DT(dtype, T, {
        hipLaunchKernelGGL((bilinear_bwd_pm_kernel<T, 12>), grid, dim3(BLK), 0, as_stream(stream), grad_out, grad_in, (int)IH, (int)IW, (int)OH,
                           (int)OW, q, rh, rw, 1.f / rh, 1.f / rw);
    });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_prelu_fwd(int dtype, const void* x, const float* slope, void* y, int64_t n, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dtype == 0 || dtype == 1, "prelu_fwd: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(n >= 0 && n % VL == 0, "prelu_fwd: element count must be a multiple of %d", VL);
    if (n == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && slope && y && al16(x) && al16(y), "prelu_fwd: null or unaligned pointer");
    const size_t units = (size_t)(n / VL);
    const unsigned blocks = (unsigned)std::min<size_t>(ceil_div((int64_t)units, BLK), 256 * 16);
    FFB6D_TRAIN_DT(dtype, T, { hipLaunchKernelGGL((prelu_fwd_kernel<T>), dim3(blocks), dim3(BLK), 0, as_stream(stream), x, slope, y, units); });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_prelu_bwd(int dtype, const void* x, const void* grad_out, const float* slope, void* grad_x, float* grad_slope,
                               int64_t n, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dtype == 0 || dtype == 1, "prelu_bwd: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(n >= 0 && n % VL == 0, "prelu_bwd: element count must be a multiple of %d", VL);
    FFB6D_REQUIRE(grad_slope, "prelu_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    FFB6D_HIP_TRY(hipMemsetAsync(grad_slope, 0, sizeof(float), st));
    if (n == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && grad_out && slope && grad_x && al16(x) && al16(grad_out) && al16(grad_x), "prelu_bwd: null or unaligned pointer");
    const size_t units = (size_t)(n / VL);
    const unsigned blocks = (unsigned)std::min<size_t>(ceil_div((int64_t)units, BLK), 256 * 16);
    FFB6D_TRAIN_DT(dtype, T, {
        hipLaunchKernelGGL((prelu_bwd_kernel<T>), dim3(blocks), dim3(BLK), 0, st, x, grad_out, slope, grad_x, grad_slope, units);
    });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
