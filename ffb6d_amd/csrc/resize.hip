// ffb6d_amd/csrc/resize.hip -- bilinear resize of NCHW float32 feature maps for gfx950.
//
// The colour branch of FFB6D up-samples seven times per forward (pspnet.py:24-28: four
// F.upsample(size=(h,w), mode='bilinear') in the pyramid-pooling module, align_corners=False;
// pspnet.py:37-42: three nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)).
// These are pure HBM streams (each writes up to 629 MB at bs=8); the stock kernel spends
// ~8 ms per forward on them.  Here one lane produces four consecutive output pixels of one
// row (a single 16-byte store), source rows stay L1/L2 resident.
//
// Arithmetic follows ATen's upsample_bilinear2d (area_pixel_compute_source_index +
// the h0lambda/h1lambda blend) operation by operation, so results agree with the reference to
// float rounding.
#include "common.h"
#include "ffb6d_ops.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

__device__ __forceinline__ float src_index(float scale, int dst, bool align_corners)
{
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

template <int V>
__global__ void __launch_bounds__(BLK)
bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int IH, int IW, int OH, int OW,
                float rh, float rw, int align_corners, size_t total /* planes*OH*OW/V */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const int owv = OW / V;
    const int xg = (int)(t % owv);
    const size_t row = t / owv;              // plane*OH + oy
    const int oy = (int)(row % OH);
    const size_t plane = row / OH;
    const float h1r = src_index(rh, oy, align_corners);
    const int h1 = (int)h1r;
    const int h1p = (h1 < IH - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float* r0 = in + (plane * IH + h1) * (size_t)IW;
    const float* r1 = r0 + (size_t)h1p * IW;
    float res[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int ox = xg * V + v;
        const float w1r = src_index(rw, ox, align_corners);
        const int w1 = (int)w1r;
        const int w1p = (w1 < IW - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        res[v] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
    }
    float* o = out + row * (size_t)OW + (size_t)xg * V;
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
    } else {
        o[0] = res[0];
    }
}

// Up-sampling fast path (horizontal scale <= 0.5, OW % 4 == 0, OH % 2 == 0): one lane produces a
// 2x4 output patch (two 16-byte stores).  Four consecutive outputs of a row read at most four
// consecutive inputs, so each source row is fetched as 4 adjacent floats once and the column
// weights/offsets are shared by both output rows: 2 loads per output instead of 4.
__global__ void __launch_bounds__(BLK)
bilinear_up_kernel(const float* __restrict__ in, float* __restrict__ out, int IH, int IW, int OH, int OW,
                   float rh, float rw, int align_corners, size_t total /* planes*(OH/2)*(OW/4) */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const int owv = OW >> 2;
    const int xg = (int)(t % owv);
    const size_t prow = t / owv;             // plane*(OH/2) + oy/2
    const int oh2 = OH >> 1;
    const int oy = (int)(prow % oh2) * 2;
    const size_t plane = prow / oh2;

    // column side, shared by both rows
    int off[4], offp[4];
    float wl[4];
    const int base = (int)src_index(rw, xg * 4, align_corners);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float w1r = src_index(rw, xg * 4 + v, align_corners);
        const int w1 = (int)w1r;
        wl[v] = w1r - (float)w1;
        off[v] = w1 - base;                                  // 0..2
        offp[v] = off[v] + ((w1 < IW - 1) ? 1 : 0);          // 0..3
    }
    int cidx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cidx[j] = min(base + j, IW - 1);

    auto pick = [](const float (&a)[4], int o) {
        return o == 0 ? a[0] : (o == 1 ? a[1] : (o == 2 ? a[2] : a[3]));
    };

    const float* pin = in + plane * (size_t)IH * IW;
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
        const float h1r = src_index(rh, oy + ry, align_corners);
        const int h1 = (int)h1r;
        const int h1p = (h1 < IH - 1) ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
        const float* r0 = pin + (size_t)h1 * IW;
        const float* r1 = r0 + (size_t)h1p * IW;
        float a[4], c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = r0[cidx[j]]; c[j] = r1[cidx[j]]; }
        float res[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float w0l = 1.f - wl[v];
            res[v] = h0l * (w0l * pick(a, off[v]) + wl[v] * pick(a, offp[v])) +
                     h1l * (w0l * pick(c, off[v]) + wl[v] * pick(c, offp[v]));
        }
        float* o = out + (plane * OH + oy + ry) * (size_t)OW + (size_t)xg * 4;
        *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
    }
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

extern "C" int ffb6d_bilinear_resize_f32(const float* in, float* out, int64_t planes, int64_t IH, int64_t IW,
                                         int64_t OH, int64_t OW, int align_corners, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(planes >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "bilinear_resize: bad shape");
    FFB6D_REQUIRE(IH < (1 << 24) && IW < (1 << 24) && OH < (1 << 24) && OW < (1 << 24), "bilinear_resize: too large");
    if (planes == 0) return FFB6D_OK;
    FFB6D_REQUIRE(in && out, "bilinear_resize: null pointer");
    // ATen: align_corners -> (in-1)/(out-1) (0 when out == 1), else in/out
    float rh, rw;
    if (align_corners) {
        rh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
        rw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    } else {
        rh = (float)IH / (float)OH;
        rw = (float)IW / (float)OW;
    }
    hipStream_t st = as_stream(stream);
    const bool vec = (OW % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec && OH % 2 == 0 && rw <= 0.5f) {
        const size_t total = (size_t)planes * (OH / 2) * (OW / 4);
        hipLaunchKernelGGL(bilinear_up_kernel, dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, st, in, out,
                           (int)IH, (int)IW, (int)OH, (int)OW, rh, rw, align_corners, total);
    } else if (vec) {
        const size_t total = (size_t)planes * OH * (OW / 4);
        hipLaunchKernelGGL((bilinear_kernel<4>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, st, in, out,
                           (int)IH, (int)IW, (int)OH, (int)OW, rh, rw, align_corners, total);
    } else {
        const size_t total = (size_t)planes * OH * OW;
        hipLaunchKernelGGL((bilinear_kernel<1>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, st, in, out,
                           (int)IH, (int)IW, (int)OH, (int)OW, rh, rw, align_corners, total);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

// ------------------------------------------------------------------------------------------
// Per-channel affine + residual + activation on NCHW maps: the eval-mode BatchNorm / ReLU / PReLU /
// residual-add glue between the colour branch's convolutions (extractors.py:49-63 BasicBlock,
// pspnet.py:34-45 PSPUpsample, ffb6d.py:30-34 stem) as ONE pass instead of up to five:
//     out = act( scale[c]*x + shift[c] + (res ? rscale[c]*res + rshift[c] : 0) )
// ------------------------------------------------------------------------------------------
namespace ffb6d {
namespace {

__global__ void __launch_bounds__(256)
affine_act_kernel(const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                  const float4* __restrict__ res, const float* __restrict__ rscale,
                  const float* __restrict__ rshift, float4* __restrict__ out, int C, int hw4, size_t total4,
                  int act, float slope)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int c = (int)((t / hw4) % C);
    const float s = scale[c], b = shift[c];
    float4 v = x[t];
    v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
    if (res) {
        const float4 r = res[t];
        const float rs = rscale ? rscale[c] : 1.f, rb = rshift ? rshift[c] : 0.f;
        v.x += r.x * rs + rb; v.y += r.y * rs + rb; v.z += r.z * rs + rb; v.w += r.w * rs + rb;
    }
    if (act == 1) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (act == 2) {
        v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
        v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
    }
    out[t] = v;
}

}  // namespace
}  // namespace ffb6d

extern "C" int ffb6d_affine_act_f32(const float* x, const float* scale, const float* shift, const float* res,
                                    const float* rscale, const float* rshift, float* out, int64_t B, int64_t C,
                                    int64_t HW, int act, float slope, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && C >= 1 && HW >= 0, "affine_act: bad shape");
    FFB6D_REQUIRE(act >= 0 && act <= 2, "affine_act: act must be 0 (none), 1 (relu) or 2 (leaky/prelu with slope)");
    if (B == 0 || HW == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && scale && shift && out, "affine_act: null pointer");
    FFB6D_REQUIRE(HW % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                                   reinterpret_cast<uintptr_t>(res)) & 15) == 0,
                  "affine_act: H*W must be a multiple of 4 and the maps 16-byte aligned");
    const size_t total4 = (size_t)B * C * (HW / 4);
    hipLaunchKernelGGL(ffb6d::affine_act_kernel, dim3((unsigned)ffb6d::ceil_div((int64_t)total4, 256)), dim3(256), 0,
                       ffb6d::as_stream(stream), reinterpret_cast<const float4*>(x), scale, shift,
                       reinterpret_cast<const float4*>(res), rscale, rshift, reinterpret_cast<float4*>(out), (int)C,
                       (int)(HW / 4), total4, act, slope);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

// ------------------------------------------------------------------------------------------
// log_softmax over the channel axis of an NCHW map (the `final` head of the colour branch,
// pspnet.py:108-112: nn.LogSoftmax() on a 4-d tensor = dim 1).  One lane owns one pixel and keeps
// its C <= 64 channel values in registers: one read + one write of the map, both coalesced.
// ------------------------------------------------------------------------------------------
namespace ffb6d {
namespace {

template <int C>
__global__ void __launch_bounds__(256)
channel_log_softmax_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, size_t total)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;   // pixel id over B*HW
    if (t >= total) return;
    const size_t b = t / HW;
    const size_t p = t - b * HW;
    const float* xi = x + b * (size_t)C * HW + p;
    float v[C];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = xi[(size_t)c * HW]; m = fmaxf(m, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s += expf(v[c] - m);
    const float lse = m + logf(s);
    float* oi = out + b * (size_t)C * HW + p;
#pragma unroll
    for (int c = 0; c < C; ++c) oi[(size_t)c * HW] = v[c] - lse;
}

}  // namespace
}  // namespace ffb6d

extern "C" int ffb6d_channel_log_softmax_f32(const float* x, float* out, int64_t B, int64_t C, int64_t HW,
                                             ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && HW >= 0, "channel_log_softmax: bad shape");
    FFB6D_REQUIRE(C == 64 || C == 32 || C == 16, "channel_log_softmax: C must be 16, 32 or 64 (got %lld)", (long long)C);
    const size_t total = (size_t)B * HW;
    if (total == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && out, "channel_log_softmax: null pointer");
    const dim3 grid((unsigned)ffb6d::ceil_div((int64_t)total, 256));
    hipStream_t st = ffb6d::as_stream(stream);
    if (C == 64) hipLaunchKernelGGL((ffb6d::channel_log_softmax_kernel<64>), grid, dim3(256), 0, st, x, out, (int)HW, total);
    else if (C == 32) hipLaunchKernelGGL((ffb6d::channel_log_softmax_kernel<32>), grid, dim3(256), 0, st, x, out, (int)HW, total);
    else hipLaunchKernelGGL((ffb6d::channel_log_softmax_kernel<16>), grid, dim3(256), 0, st, x, out, (int)HW, total);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

// ------------------------------------------------------------------------------------------
// Pyramid pooling (pspnet.py:7-31) without the 2560-channel concatenation.
//   bottleneck(cat(up(conv_i(pool_i(x))), x)) = W_x x + b + sum_i up(W_bi conv_i pool_i(x))
// (bilinear up-sampling and 1x1 convolutions are both linear and act on different axes, so they
// commute).  Two helpers:
//   psp_pool        all adaptive average pools (sizes 1,2,3,6) of a [B,C,H,W] map in ONE pass:
//                   a block stages one plane in LDS, each wave sums whole bins
//   psp_prior_sum   S[b,m,y,x] = sum_i bilinear(z_i)[b,m,y,x] for the tiny pre-multiplied maps
//                   z_i [B,M,s_i,s_i] (align_corners = False), one write of the output
// The 1x1 bottleneck itself runs on the fused shared-MLP kernel with S as its additive term.
// ------------------------------------------------------------------------------------------
namespace ffb6d {
namespace {

constexpr int PSP_MAX = 4;

struct PspSizes { int n; int s[PSP_MAX]; int off[PSP_MAX + 1]; };   // off = prefix sums of s*s

__global__ void __launch_bounds__(256)
psp_pool_kernel(const float* __restrict__ x, float* __restrict__ out /* [planes, total_bins] */, int H, int W,
                PspSizes sz)
{
    extern __shared__ float plane[];
    const size_t pl = blockIdx.x;
    const int hw = H * W;
    const float* src = x + pl * (size_t)hw;
    for (int i = threadIdx.x; i < hw; i += 256) plane[i] = src[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int total = sz.off[sz.n];
    for (int bin = wave; bin < total; bin += 4) {
        int lvl = 0;
        while (bin >= sz.off[lvl + 1]) ++lvl;
        const int s = sz.s[lvl];
        const int by = (bin - sz.off[lvl]) / s, bxi = (bin - sz.off[lvl]) % s;
        // ATen adaptive pooling: start = floor(i*in/out), end = ceil((i+1)*in/out)
        const int y0 = (by * H) / s, y1 = ((by + 1) * H + s - 1) / s;
        const int x0 = (bxi * W) / s, x1 = ((bxi + 1) * W + s - 1) / s;
        const int rw = x1 - x0, n = (y1 - y0) * rw;
        float acc = 0.f;
        for (int i = lane; i < n; i += 64) acc += plane[(y0 + i / rw) * W + x0 + i % rw];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) out[pl * total + bin] = acc / (float)n;
    }
}

__global__ void __launch_bounds__(256)
psp_prior_sum_kernel(const float* __restrict__ z /* [planes, total_bins] */, float* __restrict__ out, int H, int W,
                     PspSizes sz, size_t total4 /* planes*H*W/4 */)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int w4 = W >> 2;
    const int xg = (int)(t % w4);
    const size_t row = t / w4;
    const int oy = (int)(row % H);
    const size_t pl = row / H;
    const float* zp = z + pl * (size_t)sz.off[sz.n];
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < sz.n; ++l) {
        const int s = sz.s[l];
        const float* m = zp + sz.off[l];
        const float rh = (float)s / (float)H, rw = (float)s / (float)W;
        const float h1r = src_index(rh, oy, false);
        const int h1 = (int)h1r;
        const int h1p = (h1 < s - 1) ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float w1r = src_index(rw, xg * 4 + v, false);
            const int w1 = (int)w1r;
            const int w1p = (w1 < s - 1) ? 1 : 0;
            const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
            res[v] += h0l * (w0l * m[h1 * s + w1] + w1l * m[h1 * s + w1 + w1p]) +
                      h1l * (w0l * m[(h1 + h1p) * s + w1] + w1l * m[(h1 + h1p) * s + w1 + w1p]);
        }
    }
    *reinterpret_cast<float4*>(out + row * (size_t)W + (size_t)xg * 4) = make_float4(res[0], res[1], res[2], res[3]);
}

int fill_sizes(PspSizes& sz, const int* sizes, int n)
{
    if (n < 1 || n > PSP_MAX) return -1;
    sz.n = n;
    sz.off[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (sizes[i] < 1 || sizes[i] > 64) return -1;
        sz.s[i] = sizes[i];
        sz.off[i + 1] = sz.off[i] + sizes[i] * sizes[i];
    }
    return 0;
}

}  // namespace
}  // namespace ffb6d

extern "C" int ffb6d_psp_pool_f32(const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                                  const int* sizes, int nsizes, ffb6d_stream_t stream)
{
    ffb6d::PspSizes sz;
    FFB6D_REQUIRE(ffb6d::fill_sizes(sz, sizes, nsizes) == 0, "psp_pool: 1..4 pool sizes in [1,64] expected");
    FFB6D_REQUIRE(planes >= 0 && H >= 1 && W >= 1 && H * W * 4 <= 160 * 1024, "psp_pool: plane must fit in LDS");
    if (planes == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && out, "psp_pool: null pointer");
    hipLaunchKernelGGL(ffb6d::psp_pool_kernel, dim3((unsigned)planes), dim3(256), (size_t)(H * W * 4),
                       ffb6d::as_stream(stream), x, out, (int)H, (int)W, sz);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_psp_prior_sum_f32(const float* z, float* out, int64_t planes, int64_t H, int64_t W,
                                       const int* sizes, int nsizes, ffb6d_stream_t stream)
{
    ffb6d::PspSizes sz;
    FFB6D_REQUIRE(ffb6d::fill_sizes(sz, sizes, nsizes) == 0, "psp_prior_sum: 1..4 pool sizes in [1,64] expected");
    FFB6D_REQUIRE(planes >= 0 && H >= 1 && W >= 4 && W % 4 == 0, "psp_prior_sum: W must be a multiple of 4");
    if (planes == 0) return FFB6D_OK;
    FFB6D_REQUIRE(z && out && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "psp_prior_sum: bad pointer");
    const size_t total4 = (size_t)planes * H * (W / 4);
    hipLaunchKernelGGL(ffb6d::psp_prior_sum_kernel, dim3((unsigned)ffb6d::ceil_div((int64_t)total4, 256)), dim3(256), 0,
                       ffb6d::as_stream(stream), z, out, (int)H, (int)W, sz, total4);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

// ------------------------------------------------------------------------------------------
// Depth image -> xyz image (the dataset's `dpt_2_pcld`,
// ffb6d/datasets/linemod/linemod_dataset.py:188-199), on the device so the index pyramid can be
// built without a host round trip.  Arithmetic as numpy does it there: dpt = f32(depth)/cam_scale
// in float32, then ((col - cx) * dpt) / fx in float64 (int64 index minus float64 intrinsic), masked
// by dpt > 1e-8, NaN/Inf -> 0 (linemod_dataset.py:258-259), rounded once to float32.
// Output channel-major [B,3,H,W] (x, y, z planes), the layout build_index_pyramid consumes.
// ------------------------------------------------------------------------------------------
namespace ffb6d {
namespace {

__global__ void __launch_bounds__(256)
depth_to_cloud_kernel(const float* __restrict__ depth, const double* __restrict__ K, float cam_scale,
                      float* __restrict__ out, int H, int W, size_t total)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;   // over B*H*W
    if (t >= total) return;
    const size_t hw = (size_t)H * W;
    const size_t b = t / hw;
    const size_t pix = t - b * hw;
    const int r = (int)(pix / W), c = (int)(pix - (size_t)r * W);
    const double* k = K + b * 9;
    const float dpt = depth[t] / cam_scale;
    const float msk = dpt > 1e-8f ? 1.f : 0.f;
    double x = (((double)c - k[2]) * (double)dpt) / k[0];     // K[0][2], K[0][0]
    double y = (((double)r - k[5]) * (double)dpt) / k[4];     // K[1][2], K[1][1]
    double z = (double)dpt;
    x *= (double)msk; y *= (double)msk; z *= (double)msk;
    if (!(fabs(x) <= 1.79e308)) x = 0.0;                       // NaN / Inf -> 0
    if (!(fabs(y) <= 1.79e308)) y = 0.0;
    if (!(fabs(z) <= 1.79e308)) z = 0.0;
    float* o = out + b * 3 * hw + pix;
    o[0] = (float)x;
    o[hw] = (float)y;
    o[2 * hw] = (float)z;
}

}  // namespace
}  // namespace ffb6d

extern "C" int ffb6d_depth_to_cloud_f32(const float* depth, const double* K, float cam_scale, float* out,
                                        int64_t B, int64_t H, int64_t W, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && cam_scale != 0.f, "depth_to_cloud: bad arguments");
    const size_t total = (size_t)B * H * W;
    if (total == 0) return FFB6D_OK;
    FFB6D_REQUIRE(depth && K && out, "depth_to_cloud: null pointer");
    hipLaunchKernelGGL(ffb6d::depth_to_cloud_kernel, dim3((unsigned)ffb6d::ceil_div((int64_t)total, 256)), dim3(256), 0,
                       ffb6d::as_stream(stream), depth, K, cam_scale, out, (int)H, (int)W, total);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
