// ffb6d_amd/csrc/ops_pm.hip -- neighbour gathers, pooling and the colour-branch glue on POINT-MAJOR / PIXEL-MAJOR
// ("channels last") activations for gfx950: one row of C contiguous floats per point or pixel.
//
// Reference bodies (stock torch ops there):
//   FFB6D.random_sample          ffb6d/models/ffb6d.py:159-177   gather K rows + max over K
//   FFB6D.nearest_interpolation  ffb6d/models/ffb6d.py:179-194   1-NN row gather (also the `choose` pick, :309-312)
//   relative_pos_encoding        ffb6d/models/RandLA/RandLANet.py:216-223
//   BatchNorm/ReLU/PReLU/residual glue of the colour branch: extractors.py:49-63, pspnet.py:34-45, ffb6d.py:30-34
//   bilinear up-sampling         pspnet.py:24-28 (align_corners=False), :37-42 (align_corners=True)
//   pyramid pooling              pspnet.py:7-31
//
// In the reference's channel-major layout a gathered point touches C different 64-byte sectors (one 4-byte element
// each); here it is ONE contiguous read of 4*C bytes: lanes run along the channel axis (a float4 per lane), so every
// load and store of these kernels is a full-width coalesced access and the index is read once per row.
//
// Every kernel is templated on the element type of the rows (float, or __bf16 for the mixed-precision configuration:
// 16-byte units of 4 or 8 channels per lane, arithmetic in fp32, stores rounded to nearest even).
#include <cfloat>

#include "common.h"
#include "ffb6d_ops.h"
#include "row_unit.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

// a 16-byte unit of a row: VL consecutive channels, held as fp32 in registers (csrc/row_unit.h)
template <typename T> using Unit = RowUnit<T>;
// fp32 per-channel parameters (BatchNorm scale / shift) of a unit's VL channels
template <int VL>
__device__ __forceinline__ void load_params(const float* p, int unit, float (&out)[VL])
{
#pragma unroll
    for (int i = 0; i < VL; i += 4) {
        const float4 f = *reinterpret_cast<const float4*>(p + (size_t)unit * VL + i);
        out[i] = f.x; out[i + 1] = f.y; out[i + 2] = f.z; out[i + 3] = f.w;
    }
}

__device__ __forceinline__ float max_nan(float m, float v)   // torch.max semantics: NaN propagates
{
    return (v > m || v != v) ? v : m;
}

// ------------------------------------------------------------------------------------------------
// random_sample: out[b, n, :] = max_k F[b, idx[b, n, k], :]      (lanes = 16-byte units of a row)
// a group of q lanes owns one output point; K loads in flight per lane
// ------------------------------------------------------------------------------------------------
template <typename T, typename IdxT, int K>
__global__ void __launch_bounds__(BLK)
random_sample_pm_kernel(const void* __restrict__ feat, const IdxT* __restrict__ idx, void* __restrict__ out,
                        int q, int M, int Np, size_t total /* B*Np*q */)
{
    using U = Unit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;                 // b*Np + n
    const int c = (int)(t - pt * q);
    const size_t b = pt / Np;
    const IdxT* ip = idx + pt * K;
    const size_t base = b * (size_t)M * q + c;
    U v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = U::load(feat, base + (size_t)ip[k] * q);
    U m = v[0];
#pragma unroll
    for (int k = 1; k < K; ++k)
#pragma unroll
        for (int e = 0; e < U::VL; ++e) m.v[e] = max_nan(m.v[e], v[k].v[e]);
    m.store(out, t);
}

template <typename T, typename IdxT>
__global__ void __launch_bounds__(BLK)
random_sample_pm_anyk_kernel(const void* __restrict__ feat, const IdxT* __restrict__ idx, void* __restrict__ out,
                             int q, int M, int Np, int K, size_t total)
{
    using U = Unit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;
    const int c = (int)(t - pt * q);
    const size_t b = pt / Np;
    const IdxT* ip = idx + pt * K;
    const size_t base = b * (size_t)M * q + c;
    U m = U::load(feat, base + (size_t)ip[0] * q);
    for (int k = 1; k < K; ++k) {
        const U v = U::load(feat, base + (size_t)ip[k] * q);
#pragma unroll
        for (int e = 0; e < U::VL; ++e) m.v[e] = max_nan(m.v[e], v.v[e]);
    }
    m.store(out, t);
}

// ------------------------------------------------------------------------------------------------
// gather_rows: out[b, u, :] = F[b, idx[b, u], :]   (16-byte units are copied as they are)
// ------------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
gather_rows_pm_kernel(const uint4* __restrict__ feat, const IdxT* __restrict__ idx, uint4* __restrict__ out, int q, int M,
                      int U, size_t total /* B*U*q */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / q;                // b*U + u
    const int c = (int)(t - row * q);
    const size_t b = row / U;
    out[t] = feat[(b * M + (size_t)idx[row]) * q + c];
}

// ------------------------------------------------------------------------------------------------
// relative_pos_encoding, rows padded to 16 channels: out[b, n, k, 0:10] = [|p-q|, p-q, p, q], out[.., 10:16] = 0
// (the padding makes the row a legal K of the point-major shared MLP: lfa.mlp1's weight gets 6 zero columns).
// One lane per 16-byte unit of a pair's row (4 units in fp32, 2 in bf16): a wave stores 1 KiB contiguous per instruction.
// Arithmetic as RandLANet.py:216-223 / csrc/neighbour_ops.hip: separately rounded products and sums, IEEE sqrt (fp32).
// ------------------------------------------------------------------------------------------------
template <typename T, typename IdxT>
__global__ void __launch_bounds__(BLK)
rel_pos_enc_pm_kernel(const float* __restrict__ xyz, const IdxT* __restrict__ idx, void* __restrict__ out, int N, int K,
                      size_t total /* B*N*K*units */)
{
    using U = Unit<T>;
    constexpr int UNITS = 16 / U::VL;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pair = t / UNITS;
    const int part = (int)(t - pair * UNITS);
    const size_t pn = pair / K;              // b*N + n
    const size_t b = pn / N;
    const int j = (int)idx[pair];
    const float* p = xyz + pn * 3;
    const float* q = xyz + (b * N + j) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float qx = q[0], qy = q[1], qz = q[2];
    const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const float row[16] = {__fsqrt_rn(s), dx, dy, dz, px, py, pz, qx, qy, qz, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    U u;
#pragma unroll
    for (int e = 0; e < U::VL; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < UNITS; ++w) v = part == w ? row[w * U::VL + e] : v;
        u.v[e] = v;
    }
    u.store(out, t);
}

// ------------------------------------------------------------------------------------------------
// per-channel affine + residual + activation on [rows, C]:
//     out = act( scale[c]*x + shift[c] + (res ? rscale[c]*res + rshift[c] : 0) ),   act(v) = v > 0 ? v : slope*v
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(BLK)
affine_act_pm_kernel(const void* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                     const void* __restrict__ res, const float* __restrict__ rscale, const float* __restrict__ rshift,
                     void* __restrict__ out, int q /* units per row */, size_t total, float slope)
{
    using U = Unit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % q);
    float s[U::VL], b[U::VL];
    load_params<U::VL>(scale, c, s);
    load_params<U::VL>(shift, c, b);
    U v = U::load(x, t);
#pragma unroll
    for (int e = 0; e < U::VL; ++e) v.v[e] = v.v[e] * s[e] + b[e];
    if (res) {
        const U r = U::load(res, t);
        if (rscale) {
            float rs[U::VL], rb[U::VL];
            load_params<U::VL>(rscale, c, rs);
            load_params<U::VL>(rshift, c, rb);
#pragma unroll
            for (int e = 0; e < U::VL; ++e) v.v[e] += r.v[e] * rs[e] + rb[e];
        } else {
#pragma unroll
            for (int e = 0; e < U::VL; ++e) v.v[e] += r.v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < U::VL; ++e) v.v[e] = v.v[e] > 0.f ? v.v[e] : slope * v.v[e];     // exact PReLU for ANY learned slope (> 1, < 0)
    v.store(out, t);
}

// ------------------------------------------------------------------------------------------------
// stem of the colour branch (extractors.py: conv1 -> bn1 -> relu -> MaxPool2d(3, 2, 1), ffb6d.py:222): BatchNorm + ReLU + the
// 3x3 / stride-2 max pooling in ONE pass over the convolution's output -- the normalised map (157 MB at bs=8) is never written:
//     out[b, oy, ox, :] = max over the window's pixels INSIDE the map of relu(scale * x + shift)
// (torch pads with -inf, i.e. ignores the outside; NaN propagates like ATen's max_pool2d).  blockIdx.y = output row (b, oy),
// thread = (output column, 16-byte unit); every input pixel is read by up to four windows, out of L1 / L2.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(BLK)
affine_relu_maxpool_pm_kernel(const void* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                              void* __restrict__ out, int IH, int IW, int OH, int OW, int q)
{
    using U = Unit<T>;
    const int row = blockIdx.y;                  // b * OH + oy
    const int oy = row % OH, b = row / OH;
    const int t = blockIdx.x * BLK + threadIdx.x;
    if (t >= OW * q) return;
    const int ox = t / q, c = t - ox * q;
    float s[U::VL], sh[U::VL];
    load_params<U::VL>(scale, c, s);
    load_params<U::VL>(shift, c, sh);
    float m[U::VL];
#pragma unroll
    for (int e = 0; e < U::VL; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        if (iy < 0 || iy >= IH) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox + kx - 1;
            if (ix < 0 || ix >= IW) continue;
            const U v = U::load(x, (((size_t)b * IH + iy) * IW + ix) * q + c);
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                const float a = v.v[e] * s[e] + sh[e];
                m[e] = max_nan(m[e], a > 0.f ? a : (a == a ? 0.f : a));   // relu; NaN stays NaN
            }
        }
    }
    U o;
#pragma unroll
    for (int e = 0; e < U::VL; ++e) o.v[e] = m[e];
    o.store(out, (size_t)row * OW * q + t);
}

// ------------------------------------------------------------------------------------------------
// bilinear resize of [B, IH, IW, C] -> [B, OH, OW, C]; lane = one 16-byte unit of one output pixel;
// ATen's upsample_bilinear2d arithmetic (area_pixel_compute_source_index + the lambda blend)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float src_index(float scale, int dst, bool align_corners)
{
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

// blockIdx.y = output row (b, oy): the two source rows and the vertical weights are wave-uniform scalars;
// blockIdx.x * 256 + thread = (output column, unit) of that row
template <typename T>
__global__ void __launch_bounds__(BLK)
bilinear_pm_kernel(const void* __restrict__ in, void* __restrict__ out, int IH, int IW, int OH, int OW, int q, int qshift,
                   float rh, float rw, int align_corners)
{
    using U = Unit<T>;
    const int row = blockIdx.y;                  // b*OH + oy
    const int oy = row % OH, b = row / OH;
    const int t = blockIdx.x * BLK + threadIdx.x;
    if (t >= OW * q) return;
    const float h1r = src_index(rh, oy, align_corners);
    const int h1 = (int)h1r;
    const int h1p = (h1 < IH - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const size_t r0 = ((size_t)b * IH + h1) * IW * q;
    const size_t r1 = r0 + (size_t)h1p * IW * q;
    const int ox = qshift >= 0 ? (t >> qshift) : t / q;
    const int c = t - ox * q;
    const float w1r = src_index(rw, ox, align_corners);
    const int w1 = (int)w1r;
    const int w1p = (w1 < IW - 1) ? 1 : 0;
    const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const int i0 = w1 * q + c, i1 = i0 + w1p * q;
    const U a = U::load(in, r0 + i0), bq = U::load(in, r0 + i1), cc = U::load(in, r1 + i0), d = U::load(in, r1 + i1);
    U v;
#pragma unroll
    for (int e = 0; e < U::VL; ++e) v.v[e] = h0l * (w0l * a.v[e] + w1l * bq.v[e]) + h1l * (w0l * cc.v[e] + w1l * d.v[e]);
    v.store(out, (size_t)row * OW * q + t);
}

// The 3 x 3 patches of the bilinear (align_corners) up-sampling of `in` around picked pixels -- the operand rows of PSPUpsample's
// convolution (pspnet.py:34-45) evaluated only where its output is read: the last colour stage feeds the heads through the `choose`
// pick alone (ffb6d.py:302-312).  out[r, tap, :] = up(in)[b, Y + ky - 1, X + kx - 1, :] for idx[r] = Y * OW + X, tap = ky * 3 + kx,
// zeros where the patch leaves the OH x OW map (the convolution's zero padding).  Element arithmetic as bilinear_pm_kernel: a patch
// element IS the up-sampled map's element.  One thread per (row, filter row ky, 16-byte unit): its three taps' twelve source loads are
// issued together (taps outside the map read a clamped address and are zeroed by a select), neighbouring taps share source pixels in L1.
template <typename T, typename I>
__global__ void __launch_bounds__(BLK)
upsampled_patch_rows_pm_kernel(const void* __restrict__ in, const I* __restrict__ idx, void* __restrict__ out, int IH, int IW, int OH,
                               int OW, int q, int P, size_t total, float rh, float rw)
{
    using U = Unit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;          // (row, ky, unit)
    if (t >= total) return;
    const int c = (int)(t % (size_t)q);
    const size_t rt = t / (size_t)q;
    const int ky = (int)(rt % 3);
    const size_t r = rt / 3;
    const int b = (int)(r / (size_t)P);
    long pix = (long)idx[r];
    pix = pix < 0 ? 0 : (pix >= (long)OH * OW ? (long)OH * OW - 1 : pix);        // memory safety only: torch.gather would raise
    const int oy = (int)(pix / OW), ox = (int)(pix - (long)oy * OW);
    const int yy = oy + ky - 1;
    const bool y_in = yy >= 0 && yy < OH;
    const float h1r = src_index(rh, y_in ? yy : oy, 1);
    const int h1 = (int)h1r;
    const int h1p = (h1 < IH - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const size_t r0 = ((size_t)b * IH + h1) * IW * q;
    const size_t r1 = r0 + (size_t)h1p * IW * q;
    U a[3], bq[3], cc[3], d[3];
    float w0l[3], w1l[3];
    bool inside[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int xx = ox + kx - 1;
        const bool x_in = xx >= 0 && xx < OW;
        inside[kx] = y_in && x_in;
        const float w1r = src_index(rw, x_in ? xx : ox, 1);
        const int w1 = (int)w1r;
        const int w1p = (w1 < IW - 1) ? 1 : 0;
        w1l[kx] = w1r - (float)w1;
        w0l[kx] = 1.f - w1l[kx];
        const int i0 = w1 * q + c, i1 = i0 + w1p * q;
        a[kx] = U::load(in, r0 + i0); bq[kx] = U::load(in, r0 + i1); cc[kx] = U::load(in, r1 + i0); d[kx] = U::load(in, r1 + i1);
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        U v;
#pragma unroll
        for (int e = 0; e < U::VL; ++e) {
            const float x = h0l * (w0l[kx] * a[kx].v[e] + w1l[kx] * bq[kx].v[e]) + h1l * (w0l[kx] * cc[kx].v[e] + w1l[kx] * d[kx].v[e]);
            v.v[e] = inside[kx] ? x : 0.f;
        }
        v.store(out, ((r * 9 + (size_t)(ky * 3 + kx)) * (size_t)q) + (size_t)c);
    }
}

// ------------------------------------------------------------------------------------------------
// pyramid pooling helpers (pspnet.py:7-31 rewritten as  W_x x + b + sum_i up_i((W_b,i W_i) pool_i(x)), forward_pm.py)
//   psp_pool       all adaptive average pools of [B,H,W,C] -> fp32 [B, bins, C] (two passes through fp32 partial sums)
//   psp_prior_sum  out[b,y,x,:] = sum_levels bilinear(z_level)[b,y,x,:] for fp32 z [B, bins, M] (align_corners=False)
// ------------------------------------------------------------------------------------------------
constexpr int PSP_MAX = 4;
struct PspSizes { int n; int s[PSP_MAX]; int off[PSP_MAX + 1]; };

// Pass 1: a workgroup per (frame, image row) reads that row once and leaves, for every level, the row's partial sum of
// each horizontal bin: part[b, y, xbin, :] (fp32), xbin running over the sum(s) horizontal bins of all levels.  Pass 2: a
// workgroup per (frame, bin) adds the rows of its vertical extent.  (One workgroup per bin cannot pull the 1x1 level's
// 10 MB through a single CU in reasonable time; this way the map is streamed once by H*B workgroups.)
// Round 5: a workgroup per (frame, image row, LEVEL) -- four times the workgroups, and as many threads as the row has 16-byte units
// (512 channels = 128 units: half of a 256-thread block idled): the kernel was latency bound at 2 waves per workgroup on 480
// workgroups (80 us for 78 MB = 0.12 of HBM).  Each level re-reads the row (L2 hits after the first).
template <typename T>
__global__ void __launch_bounds__(BLK)
psp_rowsum_pm_kernel(const void* __restrict__ x, float* __restrict__ part, int H, int W, int q, int nx, PspSizes sz)
{
    using U = Unit<T>;
    const int by = blockIdx.x;                               // b*H + y
    const int l = blockIdx.y;                                // level
    const size_t src = (size_t)by * W * q;
    const int s = sz.s[l];
    int slot0 = 0;
    for (int i = 0; i < l; ++i) slot0 += sz.s[i];
    for (int c = threadIdx.x; c < q; c += blockDim.x) {
        for (int j = 0; j < s; ++j) {
            const int x0 = (j * W) / s, x1 = ((j + 1) * W + s - 1) / s;
            float acc[U::VL];
#pragma unroll
            for (int e = 0; e < U::VL; ++e) acc[e] = 0.f;
            for (int xx = x0; xx < x1; xx += 8) {          // 8 independent loads in flight, surplus ones are dropped
                U v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = U::load(x, src + (size_t)min(xx + u, x1 - 1) * q + c);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (xx + u < x1) {
#pragma unroll
                        for (int e = 0; e < U::VL; ++e) acc[e] += v[u].v[e];
                    }
            }
            float* dst = part + (((size_t)by * nx + slot0 + j) * q + c) * U::VL;
#pragma unroll
            for (int e = 0; e < U::VL; ++e) dst[e] = acc[e];
        }
    }
}

__global__ void __launch_bounds__(BLK)
psp_binsum_pm_kernel(const float* __restrict__ part, float* __restrict__ out, int H, int W, int C, int nx, PspSizes sz)
{
    const int total = sz.off[sz.n];
    const int b = blockIdx.x / total, bin = blockIdx.x % total;
    int lvl = 0, xoff = 0;
    while (bin >= sz.off[lvl + 1]) { xoff += sz.s[lvl]; ++lvl; }
    const int s = sz.s[lvl];
    const int by = (bin - sz.off[lvl]) / s, bxi = (bin - sz.off[lvl]) % s;
    // ATen adaptive pooling: start = floor(i*in/out), end = ceil((i+1)*in/out)
    const int y0 = (by * H) / s, y1 = ((by + 1) * H + s - 1) / s;
    const int x0 = (bxi * W) / s, x1 = ((bxi + 1) * W + s - 1) / s;
    const float inv = (float)((y1 - y0) * (x1 - x0));
    for (int c = threadIdx.x; c < C; c += BLK) {
        const float* src = part + ((size_t)b * H * nx + xoff + bxi) * C + c;
        float acc = 0.f;
        for (int y = y0; y < y1; ++y) acc += src[(size_t)y * nx * C];
        out[((size_t)b * total + bin) * C + c] = acc / inv;
    }
}

// general form (any row length): index arithmetic per thread
template <typename T>
__global__ void __launch_bounds__(BLK)
psp_prior_sum_pm_kernel(const float* __restrict__ z, void* __restrict__ out, int H, int W, int q, PspSizes sz,
                        size_t total /* B*H*W*q */)
{
    using U = Unit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pix = t / q;
    const int c = (int)(t - pix * q);
    const int ox = (int)(pix % W);
    const size_t row = pix / W;
    const int oy = (int)(row % H);
    const size_t b = row / H;
    const int M = q * U::VL;
    const float* zb = z + b * (size_t)sz.off[sz.n] * M + (size_t)c * U::VL;
    U res;
#pragma unroll
    for (int e = 0; e < U::VL; ++e) res.v[e] = 0.f;
    for (int l = 0; l < sz.n; ++l) {
        const int s = sz.s[l];
        const float* m = zb + (size_t)sz.off[l] * M;
        const float rh = (float)s / (float)H, rw = (float)s / (float)W;
        const float h1r = src_index(rh, oy, false);
        const int h1 = (int)h1r;
        const int h1p = (h1 < s - 1) ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
        const float w1r = src_index(rw, ox, false);
        const int w1 = (int)w1r;
        const int w1p = (w1 < s - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const float* a = m + (size_t)(h1 * s + w1) * M;
        const float* bq = m + (size_t)(h1 * s + w1 + w1p) * M;
        const float* cc = m + (size_t)((h1 + h1p) * s + w1) * M;
        const float* d = m + (size_t)((h1 + h1p) * s + w1 + w1p) * M;
#pragma unroll
        for (int e = 0; e < U::VL; ++e) res.v[e] += h0l * (w0l * a[e] + w1l * bq[e]) + h1l * (w0l * cc[e] + w1l * d[e]);
    }
    res.store(out, t);
}

// Round 5: the general form writes 157 MB (8 x 60 x 80 x 1024 fp32) in 117 us = 0.17 of HBM -- it is bound by the L2, not by HBM or the
// vector unit: every output unit re-reads its 16 corner units of z (200 KB per frame: beyond the L1), 16 bytes read per byte written,
// 21 TB/s of L2 traffic.  Here a workgroup owns XP consecutive pixels of one image row for 256 (or q) channel units: the pixel is
// UNIFORM per iteration, so source indices and weights live on the scalar unit, and a level's four corner units stay in registers until
// the pixel walk leaves the bin pair (every 13-80 pixels): z is read ~10x less often.  Same arithmetic per output element.
constexpr int PSP_XP = 16;
template <typename T>
__global__ void __launch_bounds__(BLK)
psp_prior_sum_row_pm_kernel(const float* __restrict__ z, void* __restrict__ out, int H, int W, int q, PspSizes sz)
{
    using U = Unit<T>;
    const int x_beg = blockIdx.x * PSP_XP, x_end = min(W, x_beg + PSP_XP);
    const int by = blockIdx.y;                                   // b * H + oy
    const int oy = by % H;
    const size_t b = by / H;
    const int c = blockIdx.z * blockDim.x + threadIdx.x;         // channel unit (the launcher makes q a multiple of blockDim.x)
    const int M = q * U::VL;
    const float* zb = z + b * (size_t)sz.off[sz.n] * M + (size_t)c * U::VL;
    float cor[PSP_MAX][4][U::VL];                                // corner units of every level, valid while key[l] matches
    int key[PSP_MAX];
#pragma unroll
    for (int l = 0; l < PSP_MAX; ++l) key[l] = -1;
    for (int ox = x_beg; ox < x_end; ++ox) {
        U res;
#pragma unroll
        for (int e = 0; e < U::VL; ++e) res.v[e] = 0.f;
#pragma unroll
        for (int l = 0; l < PSP_MAX; ++l) {
            if (l < sz.n) {
                const int s = sz.s[l];
                const float rh = (float)s / (float)H, rw = (float)s / (float)W;
                const float h1r = src_index(rh, oy, false);
                const int h1 = (int)h1r;
                const int h1p = (h1 < s - 1) ? 1 : 0;
                const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
                const float w1r = src_index(rw, ox, false);
                const int w1 = (int)w1r;
                const int w1p = (w1 < s - 1) ? 1 : 0;
                const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
                const int k = 2 * w1 + w1p;                      // (h1, h1p are fixed for the row)
                if (k != key[l]) {                               // uniform: the whole workgroup reloads together
                    key[l] = k;
                    const float* m = zb + (size_t)sz.off[l] * M;
                    const float* src[4] = {m + (size_t)(h1 * s + w1) * M, m + (size_t)(h1 * s + w1 + w1p) * M,
                                           m + (size_t)((h1 + h1p) * s + w1) * M, m + (size_t)((h1 + h1p) * s + w1 + w1p) * M};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < U::VL; ++e) cor[l][t][e] = src[t][e];
                }
#pragma unroll
                for (int e = 0; e < U::VL; ++e)
                    res.v[e] += h0l * (w0l * cor[l][0][e] + w1l * cor[l][1][e]) + h1l * (w0l * cor[l][2][e] + w1l * cor[l][3][e]);
            }
        }
        res.store(out, ((size_t)by * W + ox) * q + c);
    }
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

bool bits_ok(int bits) { return bits == 32 || bits == 64; }
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool dt_ok(int dtype) { return dtype == 0 || dtype == 1; }
int vl_of(int dtype) { return dtype ? 8 : 4; }

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

#define DISPATCH_IDX(bits, IdxT, ...)                       \
    if ((bits) == 64) { using IdxT = int64_t; __VA_ARGS__ } \
    else { using IdxT = int32_t; __VA_ARGS__ }
#define DISPATCH_DT(dtype, T, ...)                   \
    if ((dtype) == 1) { using T = __bf16; __VA_ARGS__ } \
    else { using T = float; __VA_ARGS__ }

extern "C" {

/* dtype of the row elements: 0 = float32, 1 = bfloat16 (channel counts must be multiples of 4 resp. 8) */

int ffb6d_random_sample_pm(int dtype, const void* feat, const void* idx, int idx_bits, void* out, int64_t B, int64_t M, int64_t C,
                           int64_t Np, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype) && bits_ok(idx_bits), "random_sample_pm: dtype must be 0/1, idx_bits 32 or 64");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(B >= 0 && C >= VL && C % VL == 0 && Np >= 0 && M >= 1 && K >= 1, "random_sample_pm: bad shape (C %% %d == 0)", VL);
    if (B == 0 || Np == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && out && al16(feat) && al16(out), "random_sample_pm: null or unaligned pointer");
    const int q = (int)(C / VL);
    const size_t total = (size_t)B * Np * q;
    const dim3 grid((unsigned)ceil_div((int64_t)total, BLK));
    hipStream_t st = as_stream(stream);
    DISPATCH_DT(dtype, T, DISPATCH_IDX(idx_bits, IdxT, {
        const IdxT* ip = static_cast<const IdxT*>(idx);
        if (K == 16)
            hipLaunchKernelGGL((random_sample_pm_kernel<T, IdxT, 16>), grid, dim3(BLK), 0, st, feat, ip, out, q, (int)M, (int)Np, total);
        else
            hipLaunchKernelGGL((random_sample_pm_anyk_kernel<T, IdxT>), grid, dim3(BLK), 0, st, feat, ip, out, q, (int)M, (int)Np, K, total);
    }))
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_gather_rows_pm(int dtype, const void* feat, const void* idx, int idx_bits, void* out, int64_t B, int64_t M, int64_t C,
                         int64_t U, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype) && bits_ok(idx_bits), "gather_rows_pm: dtype must be 0/1, idx_bits 32 or 64");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(B >= 0 && C >= VL && C % VL == 0 && U >= 0 && M >= 1, "gather_rows_pm: bad shape (C %% %d == 0)", VL);
    if (B == 0 || U == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && out && al16(feat) && al16(out), "gather_rows_pm: null or unaligned pointer");
    const int q = (int)(C / VL);
    const size_t total = (size_t)B * U * q;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((gather_rows_pm_kernel<IdxT>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0,
                           as_stream(stream), static_cast<const uint4*>(feat), static_cast<const IdxT*>(idx),
                           static_cast<uint4*>(out), q, (int)M, (int)U, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_relative_pos_encoding_pm(int dtype, const float* xyz, const void* idx, int idx_bits, void* out, int64_t B, int64_t N, int K,
                                   ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype) && bits_ok(idx_bits), "relative_pos_encoding_pm: dtype must be 0/1, idx_bits 32 or 64");
    FFB6D_REQUIRE(B >= 0 && N >= 0 && K >= 1, "relative_pos_encoding_pm: bad shape");
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(xyz && idx && out && al16(out), "relative_pos_encoding_pm: null or unaligned pointer");
    const size_t total = (size_t)B * N * K * (16 / vl_of(dtype));
    DISPATCH_DT(dtype, T, DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((rel_pos_enc_pm_kernel<T, IdxT>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0,
                           as_stream(stream), xyz, static_cast<const IdxT*>(idx), out, (int)N, K, total);
    }))
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_affine_act_pm(int dtype, const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
                        const float* rshift, void* out, int64_t rows, int64_t C, int act, float slope, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "affine_act_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(rows >= 0 && C >= VL && C % VL == 0, "affine_act_pm: C must be a positive multiple of %d", VL);
    FFB6D_REQUIRE(act >= 0 && act <= 2, "affine_act_pm: act must be 0 (none), 1 (relu) or 2 (leaky/prelu with slope)");
    if (rows == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && scale && shift && out && (rscale == nullptr) == (rshift == nullptr), "affine_act_pm: null pointer");
    FFB6D_REQUIRE(al16(x) && al16(out) && al16(res) && al16(scale) && al16(shift) && al16(rscale) && al16(rshift),
                  "affine_act_pm: 16-byte aligned pointers expected");
    const size_t total = (size_t)rows * (C / VL);
    const float sl = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
    DISPATCH_DT(dtype, T, {
        hipLaunchKernelGGL((affine_act_pm_kernel<T>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, as_stream(stream), x,
                           scale, shift, res, rscale, rshift, out, (int)(C / VL), total, sl);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_affine_relu_maxpool_pm(int dtype, const void* x, const float* scale, const float* shift, void* out, int64_t B, int64_t IH,
                                 int64_t IW, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "affine_relu_maxpool_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(B >= 0 && IH >= 1 && IW >= 1 && C >= VL && C % VL == 0, "affine_relu_maxpool_pm: bad shape");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && scale && shift && out && al16(x) && al16(out) && al16(scale) && al16(shift),
                  "affine_relu_maxpool_pm: null or unaligned pointer");
    const int64_t OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;          // floor((I + 2 - 3) / 2) + 1
    FFB6D_REQUIRE(B * OH < 65536 && IH < (1 << 24) && IW < (1 << 24), "affine_relu_maxpool_pm: too large (grid y = B * OH < 65536)");
    const int q = (int)(C / VL);
    DISPATCH_DT(dtype, T, {
        hipLaunchKernelGGL((affine_relu_maxpool_pm_kernel<T>), dim3((unsigned)ceil_div(OW * (int64_t)q, BLK), (unsigned)(B * OH)),
                           dim3(BLK), 0, as_stream(stream), x, scale, shift, out, (int)IH, (int)IW, (int)OH, (int)OW, q);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_bilinear_resize_pm(int dtype, const void* in, void* out, int64_t B, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int64_t C,
                             int align_corners, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "bilinear_resize_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(B >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1 && C >= VL && C % VL == 0, "bilinear_resize_pm: bad shape");
    FFB6D_REQUIRE(IH < (1 << 24) && IW < (1 << 24) && OH < (1 << 24) && OW < (1 << 24), "bilinear_resize_pm: too large");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(in && out && al16(in) && al16(out), "bilinear_resize_pm: null or unaligned pointer");
    float rh, rw;     // ATen: align_corners -> (in-1)/(out-1) (0 when out == 1), else in/out
    if (align_corners) {
        rh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
        rw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    } else {
        rh = (float)IH / (float)OH;
        rw = (float)IW / (float)OW;
    }
    const int q = (int)(C / VL);
    int qshift = -1;
    if ((q & (q - 1)) == 0) for (qshift = 0; (1 << qshift) < q; ++qshift) {}
    FFB6D_REQUIRE(B * OH < 65536, "bilinear_resize_pm: B * OH must stay below 65536 (grid y)");
    DISPATCH_DT(dtype, T, {
        hipLaunchKernelGGL((bilinear_pm_kernel<T>), dim3((unsigned)ceil_div(OW * (int64_t)q, BLK), (unsigned)(B * OH)), dim3(BLK), 0,
                           as_stream(stream), in, out, (int)IH, (int)IW, (int)OH, (int)OW, q, qshift, rh, rw, align_corners);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_upsampled_patch_rows_pm(int dtype, const void* in, const void* idx, int idx_bits, void* out, int64_t B, int64_t IH, int64_t IW,
                                  int64_t OH, int64_t OW, int64_t C, int64_t P, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "upsampled_patch_rows_pm: dtype must be 0 (f32) or 1 (bf16)");
    FFB6D_REQUIRE(idx_bits == 32 || idx_bits == 64, "upsampled_patch_rows_pm: idx_bits must be 32 or 64");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(B >= 0 && P >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1 && C >= VL && C % VL == 0,
                  "upsampled_patch_rows_pm: bad shape");
    FFB6D_REQUIRE(IH < (1 << 24) && IW < (1 << 24) && OH < (1 << 24) && OW < (1 << 24) && P < (1LL << 31),
                  "upsampled_patch_rows_pm: too large");
    if (B == 0 || P == 0) return FFB6D_OK;
    FFB6D_REQUIRE(in && idx && out && al16(in) && al16(out), "upsampled_patch_rows_pm: null or unaligned pointer");
    const float rh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;       // ATen area_pixel_compute_scale, align_corners
    const float rw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    const int q = (int)(C / VL);
    const size_t total = (size_t)B * P * 3 * q;              // threads: (row, filter row, 16-byte unit)
    FFB6D_REQUIRE(ceil_div((int64_t)total, BLK) < (1LL << 31), "upsampled_patch_rows_pm: too many workgroups");
    const dim3 grid((unsigned)ceil_div((int64_t)total, BLK));
    DISPATCH_DT(dtype, T, {
        if (idx_bits == 64)
            hipLaunchKernelGGL((upsampled_patch_rows_pm_kernel<T, int64_t>), grid, dim3(BLK), 0, as_stream(stream), in,
                               static_cast<const int64_t*>(idx), out, (int)IH, (int)IW, (int)OH, (int)OW, q, (int)P, total, rh, rw);
        else
            hipLaunchKernelGGL((upsampled_patch_rows_pm_kernel<T, int32_t>), grid, dim3(BLK), 0, as_stream(stream), in,
                               static_cast<const int32_t*>(idx), out, (int)IH, (int)IW, (int)OH, (int)OW, q, (int)P, total, rh, rw);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

size_t ffb6d_psp_pool_pm_workspace_bytes(int64_t B, int64_t H, int64_t C, const int* sizes, int nsizes)
{
    PspSizes sz;
    if (fill_sizes(sz, sizes, nsizes) != 0 || B <= 0 || H <= 0 || C <= 0) return 0;
    int nx = 0;
    for (int i = 0; i < sz.n; ++i) nx += sz.s[i];
    return (size_t)B * H * nx * C * sizeof(float);
}

int ffb6d_psp_pool_pm(int dtype, const void* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, const int* sizes, int nsizes,
                      void* workspace, size_t workspace_bytes, ffb6d_stream_t stream)
{
    PspSizes sz;
    FFB6D_REQUIRE(dt_ok(dtype), "psp_pool_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(fill_sizes(sz, sizes, nsizes) == 0, "psp_pool_pm: 1..4 pool sizes in [1,64] expected");
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= VL && C % VL == 0, "psp_pool_pm: bad shape");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && out && al16(x) && al16(out), "psp_pool_pm: null or unaligned pointer");
    const size_t need = ffb6d_psp_pool_pm_workspace_bytes(B, H, C, sizes, nsizes);
    if (!workspace || workspace_bytes < need || !al16(workspace))
        return set_error(FFB6D_ERR_WORKSPACE, "psp_pool_pm: 16-byte aligned workspace of %zu bytes required, got %zu", need,
                         workspace ? workspace_bytes : (size_t)0);
    int nx = 0;
    for (int i = 0; i < sz.n; ++i) nx += sz.s[i];
    const int q = (int)(C / VL);
    hipStream_t st = as_stream(stream);
    DISPATCH_DT(dtype, T, {
        const int threads = std::min(BLK, (q + 63) / 64 * 64);
        hipLaunchKernelGGL((psp_rowsum_pm_kernel<T>), dim3((unsigned)(B * H), (unsigned)sz.n), dim3(threads), 0, st, x,
                           static_cast<float*>(workspace), (int)H, (int)W, q, nx, sz);
    })
    hipLaunchKernelGGL(psp_binsum_pm_kernel, dim3((unsigned)(B * sz.off[sz.n])), dim3(BLK), 0, st,
                       static_cast<const float*>(workspace), out, (int)H, (int)W, (int)C, nx, sz);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_psp_prior_sum_pm(int dtype, const float* z, void* out, int64_t B, int64_t H, int64_t W, int64_t M, const int* sizes,
                           int nsizes, ffb6d_stream_t stream)
{
    PspSizes sz;
    FFB6D_REQUIRE(dt_ok(dtype), "psp_prior_sum_pm: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = vl_of(dtype);
    FFB6D_REQUIRE(fill_sizes(sz, sizes, nsizes) == 0, "psp_prior_sum_pm: 1..4 pool sizes in [1,64] expected");
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && M >= VL && M % VL == 0, "psp_prior_sum_pm: bad shape");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(z && out && al16(out), "psp_prior_sum_pm: bad pointer");
    const int q = (int)(M / VL);
    const size_t npix = (size_t)B * H * W;
    if (q % 64 == 0 && (q <= BLK || q % BLK == 0) && B * H < 65536) {      // whole waves per pixel: a workgroup walks XP pixels of a row
        const int threads = std::min(q, BLK);
        const dim3 grid((unsigned)ceil_div(W, PSP_XP), (unsigned)(B * H), (unsigned)(q / threads));
        DISPATCH_DT(dtype, T, {
            hipLaunchKernelGGL((psp_prior_sum_row_pm_kernel<T>), grid, dim3(threads), 0, as_stream(stream), z, out, (int)H, (int)W, q, sz);
        })
    } else {
        const size_t total = npix * q;
        DISPATCH_DT(dtype, T, {
            hipLaunchKernelGGL((psp_prior_sum_pm_kernel<T>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, as_stream(stream),
                               z, out, (int)H, (int)W, q, sz, total);
        })
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // extern "C"
