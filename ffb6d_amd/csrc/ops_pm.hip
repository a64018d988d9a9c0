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
#include <cfloat>

#include "common.h"
#include "ffb6d_ops.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

__device__ __forceinline__ float max_nan(float m, float v)   // torch.max semantics: NaN propagates
{
    return (v > m || v != v) ? v : m;
}

// ------------------------------------------------------------------------------------------------
// random_sample: out[b, n, :] = max_k F[b, idx[b, n, k], :]      (C % 4 == 0; lanes = float4 of a row)
// a row of q = C/4 lanes owns one output point; K loads in flight per lane
// ------------------------------------------------------------------------------------------------
template <typename IdxT, int K>
__global__ void __launch_bounds__(BLK)
random_sample_pm_kernel(const float4* __restrict__ feat, const IdxT* __restrict__ idx, float4* __restrict__ out,
                        int q /* C/4 */, int M, int Np, size_t total /* B*Np*q */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;                 // b*Np + n
    const int c4 = (int)(t - pt * q);
    const size_t b = pt / Np;
    const IdxT* ip = idx + pt * K;
    const float4* base = feat + b * (size_t)M * q + c4;
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = base[(size_t)ip[k] * q];
    float4 m = v[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
        m.x = max_nan(m.x, v[k].x); m.y = max_nan(m.y, v[k].y); m.z = max_nan(m.z, v[k].z); m.w = max_nan(m.w, v[k].w);
    }
    out[t] = m;
}

template <typename IdxT>
__global__ void __launch_bounds__(BLK)
random_sample_pm_anyk_kernel(const float4* __restrict__ feat, const IdxT* __restrict__ idx, float4* __restrict__ out,
                             int q, int M, int Np, int K, size_t total)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;
    const int c4 = (int)(t - pt * q);
    const size_t b = pt / Np;
    const IdxT* ip = idx + pt * K;
    const float4* base = feat + b * (size_t)M * q + c4;
    float4 m = base[(size_t)ip[0] * q];
    for (int k = 1; k < K; ++k) {
        const float4 v = base[(size_t)ip[k] * q];
        m.x = max_nan(m.x, v.x); m.y = max_nan(m.y, v.y); m.z = max_nan(m.z, v.z); m.w = max_nan(m.w, v.w);
    }
    out[t] = m;
}

// ------------------------------------------------------------------------------------------------
// gather_rows: out[b, u, :] = F[b, idx[b, u], :]
// ------------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
gather_rows_pm_kernel(const float4* __restrict__ feat, const IdxT* __restrict__ idx, float4* __restrict__ out, int q, int M,
                      int U, size_t total /* B*U*q */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / q;                // b*U + u
    const int c4 = (int)(t - row * q);
    const size_t b = row / U;
    out[t] = feat[(b * M + (size_t)idx[row]) * q + c4];
}

// ------------------------------------------------------------------------------------------------
// relative_pos_encoding, rows padded to 16 floats: out[b, n, k, 0:10] = [|p-q|, p-q, p, q], out[.., 10:16] = 0
// (the padding makes the row a legal K of the point-major shared MLP: lfa.mlp1's weight gets 6 zero columns).
// Four lanes per (n, k) pair, one float4 each: a wave stores 1 KiB contiguous per instruction.
// Arithmetic as RandLANet.py:216-223 / csrc/neighbour_ops.hip: separately rounded products and sums, IEEE sqrt.
// ------------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
rel_pos_enc_pm_kernel(const float* __restrict__ xyz, const IdxT* __restrict__ idx, float4* __restrict__ out, int N, int K,
                      size_t total /* B*N*K*4 */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pair = t >> 2;
    const int part = (int)(t & 3);
    const size_t pn = pair / K;              // b*N + n
    const size_t b = pn / N;
    const int j = (int)idx[pair];
    const float* p = xyz + pn * 3;
    const float* q = xyz + (b * N + j) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float qx = q[0], qy = q[1], qz = q[2];
    const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
    float4 v;
    if (part == 0) {
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        v = make_float4(__fsqrt_rn(s), dx, dy, dz);
    } else if (part == 1) {
        v = make_float4(px, py, pz, qx);
    } else if (part == 2) {
        v = make_float4(qy, qz, 0.f, 0.f);
    } else {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    out[t] = v;
}

// ------------------------------------------------------------------------------------------------
// per-channel affine + residual + activation on [rows, C]:
//     out = act( scale[c]*x + shift[c] + (res ? rscale[c]*res + rshift[c] : 0) ),   act(v) = max(v, slope*v)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK)
affine_act_pm_kernel(const float4* __restrict__ x, const float4* __restrict__ scale, const float4* __restrict__ shift,
                     const float4* __restrict__ res, const float4* __restrict__ rscale, const float4* __restrict__ rshift,
                     float4* __restrict__ out, int q /* C/4 */, size_t total4, float slope)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total4) return;
    const int c4 = (int)(t % q);
    const float4 s = scale[c4], b = shift[c4];
    float4 v = x[t];
    v.x = v.x * s.x + b.x; v.y = v.y * s.y + b.y; v.z = v.z * s.z + b.z; v.w = v.w * s.w + b.w;
    if (res) {
        const float4 r = res[t];
        if (rscale) {
            const float4 rs = rscale[c4], rb = rshift[c4];
            v.x += r.x * rs.x + rb.x; v.y += r.y * rs.y + rb.y; v.z += r.z * rs.z + rb.z; v.w += r.w * rs.w + rb.w;
        } else {
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
    }
    out[t] = make_float4(fmaxf(v.x, slope * v.x), fmaxf(v.y, slope * v.y), fmaxf(v.z, slope * v.z), fmaxf(v.w, slope * v.w));
}

// ------------------------------------------------------------------------------------------------
// bilinear resize of [B, IH, IW, C] -> [B, OH, OW, C]; lane = float4 of channels of one output pixel;
// ATen's upsample_bilinear2d arithmetic (area_pixel_compute_source_index + the lambda blend), as csrc/resize.hip
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float src_index(float scale, int dst, bool align_corners)
{
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

// blockIdx.y = output row (b, oy): the two source rows and the vertical weights are wave-uniform scalars;
// blockIdx.x * 256 + thread = (output column, channel quad) of that row, one float4 per thread
__global__ void __launch_bounds__(BLK)
bilinear_pm_kernel(const float4* __restrict__ in, float4* __restrict__ out, int IH, int IW, int OH, int OW, int q, int qshift,
                   float rh, float rw, int align_corners)
{
    const int row = blockIdx.y;                  // b*OH + oy
    const int oy = row % OH, b = row / OH;
    const int t = blockIdx.x * BLK + threadIdx.x;
    if (t >= OW * q) return;
    const float h1r = src_index(rh, oy, align_corners);
    const int h1 = (int)h1r;
    const int h1p = (h1 < IH - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float4* r0 = in + ((size_t)b * IH + h1) * IW * q;
    const float4* r1 = r0 + (size_t)h1p * IW * q;
    const int ox = qshift >= 0 ? (t >> qshift) : t / q;
    const int c4 = t - ox * q;
    const float w1r = src_index(rw, ox, align_corners);
    const int w1 = (int)w1r;
    const int w1p = (w1 < IW - 1) ? 1 : 0;
    const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const int i0 = w1 * q + c4, i1 = i0 + w1p * q;
    const float4 a = r0[i0], bq = r0[i1], c = r1[i0], d = r1[i1];
    float4 v;
    v.x = h0l * (w0l * a.x + w1l * bq.x) + h1l * (w0l * c.x + w1l * d.x);
    v.y = h0l * (w0l * a.y + w1l * bq.y) + h1l * (w0l * c.y + w1l * d.y);
    v.z = h0l * (w0l * a.z + w1l * bq.z) + h1l * (w0l * c.z + w1l * d.z);
    v.w = h0l * (w0l * a.w + w1l * bq.w) + h1l * (w0l * c.w + w1l * d.w);
    out[(size_t)row * OW * q + t] = v;
}

// ------------------------------------------------------------------------------------------------
// pyramid pooling helpers (pspnet.py:7-31 rewritten as  W_x x + b + sum_i up_i((W_b,i W_i) pool_i(x)), model.py)
//   psp_pool       all adaptive average pools of [B,H,W,C] in one launch -> [B, bins, C]: a workgroup per (frame, bin),
//                  lanes = float4 of channels x pixel groups, partial sums meet in LDS
//   psp_prior_sum  out[b,y,x,:] = sum_levels bilinear(z_level)[b,y,x,:] for z [B, bins, M] (align_corners=False)
// ------------------------------------------------------------------------------------------------
constexpr int PSP_MAX = 4;
struct PspSizes { int n; int s[PSP_MAX]; int off[PSP_MAX + 1]; };

// Pass 1: a workgroup per (frame, image row) reads that row once and leaves, for every level, the row's partial sum of
// each horizontal bin: part[b, y, xbin, :], xbin running over the sum(s) horizontal bins of all levels.  Pass 2: a
// workgroup per (frame, bin) adds the rows of its vertical extent.  (One workgroup per bin cannot pull the 1x1 level's
// 10 MB through a single CU in reasonable time; this way the map is streamed once by H*B workgroups.)
__global__ void __launch_bounds__(BLK)
psp_rowsum_pm_kernel(const float4* __restrict__ x, float4* __restrict__ part, int H, int W, int q, int nx, PspSizes sz)
{
    const int by = blockIdx.x;                               // b*H + y
    const float4* src = x + (size_t)by * W * q;
    for (int c4 = threadIdx.x; c4 < q; c4 += BLK) {
        int slot = 0;
        for (int l = 0; l < sz.n; ++l) {
            const int s = sz.s[l];
            for (int j = 0; j < s; ++j, ++slot) {
                const int x0 = (j * W) / s, x1 = ((j + 1) * W + s - 1) / s;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int xx = x0; xx < x1; xx += 8) {          // 8 independent loads in flight, surplus ones are dropped
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(xx + u, x1 - 1) * q + c4];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (xx + u < x1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
                part[((size_t)by * nx + slot) * q + c4] = acc;
            }
        }
    }
}

__global__ void __launch_bounds__(BLK)
psp_binsum_pm_kernel(const float4* __restrict__ part, float4* __restrict__ out, int H, int W, int q, int nx, PspSizes sz)
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
    for (int c4 = threadIdx.x; c4 < q; c4 += BLK) {
        const float4* src = part + ((size_t)b * H * nx + xoff + bxi) * q + c4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int y = y0; y < y1; ++y) {
            const float4 v = src[(size_t)y * nx * q];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        out[((size_t)b * total + bin) * q + c4] = make_float4(acc.x / inv, acc.y / inv, acc.z / inv, acc.w / inv);
    }
}

__global__ void __launch_bounds__(BLK)
psp_prior_sum_pm_kernel(const float4* __restrict__ z, float4* __restrict__ out, int H, int W, int q, PspSizes sz,
                        size_t total /* B*H*W*q */)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pix = t / q;
    const int c4 = (int)(t - pix * q);
    const int ox = (int)(pix % W);
    const size_t row = pix / W;
    const int oy = (int)(row % H);
    const size_t b = row / H;
    const float4* zb = z + b * (size_t)sz.off[sz.n] * q + c4;
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < sz.n; ++l) {
        const int s = sz.s[l];
        const float4* m = zb + (size_t)sz.off[l] * q;
        const float rh = (float)s / (float)H, rw = (float)s / (float)W;
        const float h1r = src_index(rh, oy, false);
        const int h1 = (int)h1r;
        const int h1p = (h1 < s - 1) ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
        const float w1r = src_index(rw, ox, false);
        const int w1 = (int)w1r;
        const int w1p = (w1 < s - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const float4 a = m[(size_t)(h1 * s + w1) * q], bq = m[(size_t)(h1 * s + w1 + w1p) * q];
        const float4 c = m[(size_t)((h1 + h1p) * s + w1) * q], d = m[(size_t)((h1 + h1p) * s + w1 + w1p) * q];
        res.x += h0l * (w0l * a.x + w1l * bq.x) + h1l * (w0l * c.x + w1l * d.x);
        res.y += h0l * (w0l * a.y + w1l * bq.y) + h1l * (w0l * c.y + w1l * d.y);
        res.z += h0l * (w0l * a.z + w1l * bq.z) + h1l * (w0l * c.z + w1l * d.z);
        res.w += h0l * (w0l * a.w + w1l * bq.w) + h1l * (w0l * c.w + w1l * d.w);
    }
    out[t] = res;
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

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

#define DISPATCH_IDX(bits, IdxT, ...)                       \
    if ((bits) == 64) { using IdxT = int64_t; __VA_ARGS__ } \
    else { using IdxT = int32_t; __VA_ARGS__ }

extern "C" {

int ffb6d_random_sample_pm_f32(const float* feat, const void* idx, int idx_bits, float* out, int64_t B, int64_t M, int64_t C,
                               int64_t Np, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "random_sample_pm: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 4 && (C & 3) == 0 && Np >= 0 && M >= 1 && K >= 1, "random_sample_pm: bad shape (C %% 4 == 0)");
    if (B == 0 || Np == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && out && al16(feat) && al16(out), "random_sample_pm: null or unaligned pointer");
    const int q = (int)(C / 4);
    const size_t total = (size_t)B * Np * q;
    const dim3 grid((unsigned)ceil_div((int64_t)total, BLK));
    hipStream_t st = as_stream(stream);
    const float4* f4 = reinterpret_cast<const float4*>(feat);
    float4* o4 = reinterpret_cast<float4*>(out);
    DISPATCH_IDX(idx_bits, IdxT, {
        const IdxT* ip = static_cast<const IdxT*>(idx);
        if (K == 16)
            hipLaunchKernelGGL((random_sample_pm_kernel<IdxT, 16>), grid, dim3(BLK), 0, st, f4, ip, o4, q, (int)M, (int)Np, total);
        else
            hipLaunchKernelGGL((random_sample_pm_anyk_kernel<IdxT>), grid, dim3(BLK), 0, st, f4, ip, o4, q, (int)M, (int)Np, K, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_gather_rows_pm_f32(const float* feat, const void* idx, int idx_bits, float* out, int64_t B, int64_t M, int64_t C,
                             int64_t U, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "gather_rows_pm: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 4 && (C & 3) == 0 && U >= 0 && M >= 1, "gather_rows_pm: bad shape (C %% 4 == 0)");
    if (B == 0 || U == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && out && al16(feat) && al16(out), "gather_rows_pm: null or unaligned pointer");
    const int q = (int)(C / 4);
    const size_t total = (size_t)B * U * q;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((gather_rows_pm_kernel<IdxT>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0,
                           as_stream(stream), reinterpret_cast<const float4*>(feat), static_cast<const IdxT*>(idx),
                           reinterpret_cast<float4*>(out), q, (int)M, (int)U, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_relative_pos_encoding_pm_f32(const float* xyz, const void* idx, int idx_bits, float* out, int64_t B, int64_t N, int K,
                                       ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "relative_pos_encoding_pm: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && N >= 0 && K >= 1, "relative_pos_encoding_pm: bad shape");
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(xyz && idx && out && al16(out), "relative_pos_encoding_pm: null or unaligned pointer");
    const size_t total = (size_t)B * N * K * 4;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((rel_pos_enc_pm_kernel<IdxT>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0,
                           as_stream(stream), xyz, static_cast<const IdxT*>(idx), reinterpret_cast<float4*>(out), (int)N, K, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_affine_act_pm_f32(const float* x, const float* scale, const float* shift, const float* res, const float* rscale,
                            const float* rshift, float* out, int64_t rows, int64_t C, int act, float slope, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(rows >= 0 && C >= 4 && (C & 3) == 0, "affine_act_pm: C must be a positive multiple of 4");
    FFB6D_REQUIRE(act >= 0 && act <= 2, "affine_act_pm: act must be 0 (none), 1 (relu) or 2 (leaky/prelu with slope)");
    if (rows == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && scale && shift && out && (rscale == nullptr) == (rshift == nullptr), "affine_act_pm: null pointer");
    FFB6D_REQUIRE(al16(x) && al16(out) && al16(res) && al16(scale) && al16(shift) && al16(rscale) && al16(rshift),
                  "affine_act_pm: 16-byte aligned pointers expected");
    const size_t total4 = (size_t)rows * (C / 4);
    const float sl = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
    auto f4 = [](const float* p) { return reinterpret_cast<const float4*>(p); };
    hipLaunchKernelGGL(affine_act_pm_kernel, dim3((unsigned)ceil_div((int64_t)total4, BLK)), dim3(BLK), 0, as_stream(stream), f4(x),
                       f4(scale), f4(shift), f4(res), f4(rscale), f4(rshift), reinterpret_cast<float4*>(out), (int)(C / 4), total4, sl);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_bilinear_resize_pm_f32(const float* in, float* out, int64_t B, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int64_t C,
                                 int align_corners, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1 && C >= 4 && (C & 3) == 0, "bilinear_resize_pm: bad shape");
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
    FFB6D_REQUIRE(B * OH < (1LL << 31) && OW * (C / 4) < (1LL << 31), "bilinear_resize_pm: too large");
    const int q = (int)(C / 4);
    int qshift = -1;
    if ((q & (q - 1)) == 0) for (qshift = 0; (1 << qshift) < q; ++qshift) {}
    FFB6D_REQUIRE(B * OH < 65536, "bilinear_resize_pm: B * OH must stay below 65536 (grid y)");
    hipLaunchKernelGGL(bilinear_pm_kernel, dim3((unsigned)ceil_div(OW * (int64_t)q, BLK), (unsigned)(B * OH)), dim3(BLK), 0,
                       as_stream(stream), reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), (int)IH, (int)IW,
                       (int)OH, (int)OW, q, qshift, rh, rw, align_corners);
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

int ffb6d_psp_pool_pm_f32(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, const int* sizes, int nsizes,
                          void* workspace, size_t workspace_bytes, ffb6d_stream_t stream)
{
    PspSizes sz;
    FFB6D_REQUIRE(fill_sizes(sz, sizes, nsizes) == 0, "psp_pool_pm: 1..4 pool sizes in [1,64] expected");
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= 4 && (C & 3) == 0, "psp_pool_pm: bad shape");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && out && al16(x) && al16(out), "psp_pool_pm: null or unaligned pointer");
    const size_t need = ffb6d_psp_pool_pm_workspace_bytes(B, H, C, sizes, nsizes);
    if (!workspace || workspace_bytes < need || !al16(workspace))
        return set_error(FFB6D_ERR_WORKSPACE, "psp_pool_pm: 16-byte aligned workspace of %zu bytes required, got %zu", need,
                         workspace ? workspace_bytes : (size_t)0);
    int nx = 0;
    for (int i = 0; i < sz.n; ++i) nx += sz.s[i];
    const int q = (int)(C / 4);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(psp_rowsum_pm_kernel, dim3((unsigned)(B * H)), dim3(BLK), 0, st, reinterpret_cast<const float4*>(x),
                       static_cast<float4*>(workspace), (int)H, (int)W, q, nx, sz);
    hipLaunchKernelGGL(psp_binsum_pm_kernel, dim3((unsigned)(B * sz.off[sz.n])), dim3(BLK), 0, st,
                       static_cast<const float4*>(workspace), reinterpret_cast<float4*>(out), (int)H, (int)W, q, nx, sz);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_psp_prior_sum_pm_f32(const float* z, float* out, int64_t B, int64_t H, int64_t W, int64_t M, const int* sizes, int nsizes,
                               ffb6d_stream_t stream)
{
    PspSizes sz;
    FFB6D_REQUIRE(fill_sizes(sz, sizes, nsizes) == 0, "psp_prior_sum_pm: 1..4 pool sizes in [1,64] expected");
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && M >= 4 && (M & 3) == 0, "psp_prior_sum_pm: bad shape");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(z && out && al16(z) && al16(out), "psp_prior_sum_pm: null or unaligned pointer");
    const size_t total = (size_t)B * H * W * (M / 4);
    hipLaunchKernelGGL(psp_prior_sum_pm_kernel, dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(z), reinterpret_cast<float4*>(out), (int)H, (int)W, (int)(M / 4), sz, total);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // extern "C"
