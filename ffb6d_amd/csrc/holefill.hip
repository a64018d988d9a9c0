// ffb6d_amd/csrc/holefill.hip -- depth hole filling of the YCB input pipeline on gfx950.
//
// Reference: Basic_Utils.fill_missing (ffb6d/utils/basic_utils.py:467-487, called at datasets/ycb/ycb_dataset.py:204 as
// fill_missing(dpt_um, cam_scale, 1)) -> depth_map_utils.fill_in_multiscale(depth, extrapolate=False, blur_type='bilateral',
// max_depth=3.0) (ffb6d/utils/ip_basic/ip_basic/depth_map_utils_ycb.py:290-445, the vendored IP-Basic completion).  That
// function is a chain of OpenCV image operators; each kernel below restates one link with OpenCV's documented semantics:
//   cv2.dilate / erode      max / min over the kernel footprint, anchor at the centre, pixels outside the image never win
//                           (morphologyDefaultBorderValue)
//   cv2.morphologyEx CLOSE  dilate, then erode, same kernel
//   cv2.medianBlur(5)       median of the 5x5 window, BORDER_REPLICATE
//   cv2.bilateralFilter(src, 5, 0.5, 2.0)   radius 2, the 13 taps with i*i + j*j <= 4, BORDER_REFLECT_101, weight
//                           exp(-r^2 / (2 sigma_s^2)) * exp(-(v - v0)^2 / (2 sigma_c^2)); OpenCV evaluates the colour term
//                           through a 4096-bin interpolated table, here it is the exact exponential
// PARITY UNPINNED: OpenCV is not available in this image, so neither these kernels nor the CPU restatement
// (oracle/holefill_ref.py, numpy + scipy.ndimage) can be compared with the reference's output; they are compared with
// each other, step by step.  One thread per pixel, frames batched in blockIdx.y; images are tiny (1.2 MB per frame).
#include <cfloat>

#include "common.h"
#include "ffb6d_ops.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;
constexpr float EPS = 0.01f;        // the "valid depth" threshold used throughout depth_map_utils_ycb.py

struct Pix {
    int b, y, x;
    bool ok;
};
__device__ __forceinline__ Pix pixel(int H, int W)
{
    Pix p;
    p.b = blockIdx.y;
    const int i = blockIdx.x * BLK + threadIdx.x;
    p.ok = i < H * W;
    p.y = i / W;
    p.x = i - p.y * W;
    return p;
}

// :313-352  bin masks, inversion, the three masked cross dilations (3 / 5 / 7) and their combination, far -> near
__global__ void __launch_bounds__(BLK)
hf_multiscale_kernel(const float* __restrict__ depth, float* __restrict__ s1_out, float* __restrict__ s2_out, int H, int W,
                     float max_depth, double cam_scale, double scale_2_80m)
{
    const Pix p = pixel(H, W);
    if (!p.ok) return;
    const float* img = depth + (size_t)p.b * H * W;
    // basic_utils.py:468 `dpt / cam_scale * scale_2_80m` runs in float64 (integer image, Python floats); :310 casts to float32
    auto to_metres = [&](float v) { return (float)((double)v / cam_scale * scale_2_80m); };
    auto s1 = [&](float d) { return d > EPS ? max_depth - d : d; };
    float far_ = -FLT_MAX, med = -FLT_MAX, near_ = -FLT_MAX;
    auto tap = [&](int dy, int dx) {
        const int yy = p.y + dy, xx = p.x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) return;
        const float d = to_metres(img[(size_t)yy * W + xx]);
        const float v = s1(d);
        const int r = max(abs(dy), abs(dx));
        // np.multiply(s1, mask): masked-out pixels contribute 0
        if (r <= 1) far_ = fmaxf(far_, d > 2.0f ? v : 0.f);
        if (r <= 2) med = fmaxf(med, (d > 1.0f && d <= 2.0f) ? v : 0.f);
        near_ = fmaxf(near_, (d > EPS && d <= 1.0f) ? v : 0.f);
    };
    tap(0, 0);
#pragma unroll
    for (int k = 1; k <= 3; ++k) { tap(-k, 0); tap(k, 0); tap(0, -k); tap(0, k); }      // cross kernels 3 / 5 / 7
    const float d0 = to_metres(img[(size_t)p.y * W + p.x]);
    float v = s1(d0);
    s1_out[(size_t)p.b * H * W + (size_t)p.y * W + p.x] = v;
    if (far_ > EPS) v = far_;
    if (med > EPS) v = med;
    if (near_ > EPS) v = near_;
    s2_out[(size_t)p.b * H * W + (size_t)p.y * W + p.x] = v;
}

// full k x k dilation (DILATE) or erosion; FILL: write the dilated value only where the pixel is empty and at or below the
// column's top row (:370-378 `empty_pixels = ~valid & top_mask`, cmp_le; :398-401 `(s7 < 0.01) & top_mask`, strict)
template <bool ERODE>
__global__ void __launch_bounds__(BLK)
hf_morph_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ top_row, int H, int W, int k,
                int fill_mode /* 0 = plain, 1 = fill where v <= EPS, 2 = fill where v < EPS */)
{
    const Pix p = pixel(H, W);
    if (!p.ok) return;
    const float* img = src + (size_t)p.b * H * W;
    const float v0 = img[(size_t)p.y * W + p.x];
    float out = v0;
    bool need = true;
    if (fill_mode) {
        const bool empty = fill_mode == 1 ? !(v0 > EPS) : (v0 < EPS);
        need = empty && p.y >= top_row[(size_t)p.b * W + p.x];
    }
    if (need) {
        const int r = k / 2;
        float m = ERODE ? FLT_MAX : -FLT_MAX;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = p.y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = p.x + dx;
                if (xx < 0 || xx >= W) continue;
                const float v = img[(size_t)yy * W + xx];
                m = ERODE ? fminf(m, v) : fmaxf(m, v);
            }
        }
        out = m;
    }
    dst[(size_t)p.b * H * W + (size_t)p.y * W + p.x] = out;
}

// cv2.medianBlur(src, 5) applied where (gate > EPS) [and y >= top_row]: :360-363 and :404-406
__global__ void __launch_bounds__(BLK)
hf_median_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ top_row, int H, int W)
{
    const Pix p = pixel(H, W);
    if (!p.ok) return;
    const float* img = src + (size_t)p.b * H * W;
    const float v0 = img[(size_t)p.y * W + p.x];
    float out = v0;
    const bool gate = v0 > EPS && (!top_row || p.y >= top_row[(size_t)p.b * W + p.x]);
    if (gate) {
        float v[25];
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx)
                v[(dy + 2) * 5 + dx + 2] = img[(size_t)min(max(p.y + dy, 0), H - 1) * W + min(max(p.x + dx, 0), W - 1)];   // REPLICATE
        // 13 rounds of selection: the 13th smallest of 25
#pragma unroll
        for (int i = 0; i < 13; ++i) {
#pragma unroll
            for (int j = i + 1; j < 25; ++j) {
                const float lo = fminf(v[i], v[j]), hi = fmaxf(v[i], v[j]);
                v[i] = lo;
                v[j] = hi;
            }
        }
        out = v[12];
    }
    dst[(size_t)p.b * H * W + (size_t)p.y * W + p.x] = out;
}

// np.argmax(column > 0.01): first valid row of every column, 0 when the column has none (:366-369, :384)
__global__ void __launch_bounds__(BLK)
hf_top_row_kernel(const float* __restrict__ src, int* __restrict__ top_row, int H, int W)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x * BLK + threadIdx.x;
    if (x >= W) return;
    const float* img = src + (size_t)b * H * W;
    int row = 0;
    for (int y = 0; y < H; ++y)
        if (img[(size_t)y * W + x] > EPS) { row = y; break; }
    top_row[(size_t)b * W + x] = row;
}

// :413-417 bilateral blur, written where the PRE-median image was valid and below the top row (`valid_pixels` of :405 is
// reused), then :420-423 the final inversion, and fill_missing's rescaling (basic_utils.py:485)
__global__ void __launch_bounds__(BLK)
hf_bilateral_invert_kernel(const float* __restrict__ src, const float* __restrict__ gate_img, const int* __restrict__ top_row,
                           float* __restrict__ dst, int H, int W, float sigma_color, float sigma_space, float max_depth,
                           float cam_scale, float scale_2_80m)
{
    const Pix p = pixel(H, W);
    if (!p.ok) return;
    const float* img = src + (size_t)p.b * H * W;
    const size_t o = (size_t)p.b * H * W + (size_t)p.y * W + p.x;
    const float v0 = img[(size_t)p.y * W + p.x];
    float out = v0;
    if (gate_img[o] > EPS && p.y >= top_row[(size_t)p.b * W + p.x]) {
        const float cc = -0.5f / (sigma_color * sigma_color), cs = -0.5f / (sigma_space * sigma_space);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                if (dy * dy + dx * dx > 4) continue;
                int yy = p.y + dy, xx = p.x + dx;                 // BORDER_REFLECT_101
                yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
                xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
                const float v = img[(size_t)yy * W + xx];
                const float w = expf((float)(dy * dy + dx * dx) * cs) * expf((v - v0) * (v - v0) * cc);
                num += w * v;
                den += w;
            }
        out = num / den;
    }
    if (out > EPS) out = max_depth - out;
    dst[o] = out / scale_2_80m * cam_scale;            // basic_utils.py:485, float32 array with Python scalars
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

extern "C" size_t ffb6d_fill_missing_workspace_bytes(int64_t B, int64_t H, int64_t W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * H * W * sizeof(float) * 3 + (size_t)B * W * sizeof(int) + 256;
}

extern "C" int ffb6d_fill_missing_f32(const float* depth, double cam_scale, double scale_2_80m, float max_depth, float* out,
                                      int64_t B, int64_t H, int64_t W, void* workspace, size_t workspace_bytes,
                                      ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && H >= 5 && W >= 5 && H * W < (1LL << 31) && B < 65536, "fill_missing: bad shape (H, W >= 5)");
    FFB6D_REQUIRE(cam_scale != 0. && scale_2_80m != 0., "fill_missing: zero scale");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(depth && out, "fill_missing: null pointer");
    const size_t need = ffb6d_fill_missing_workspace_bytes(B, H, W);
    if (!workspace || workspace_bytes < need)
        return set_error(FFB6D_ERR_WORKSPACE, "fill_missing: workspace of %zu bytes required, got %zu", need,
                         workspace ? workspace_bytes : (size_t)0);
    const size_t n = (size_t)B * H * W;
    float* a = static_cast<float*>(workspace);
    float* b = a + n;
    float* c = b + n;
    int* top = reinterpret_cast<int*>(c + n);
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)ceil_div(H * W, BLK), (unsigned)B), cgrid((unsigned)ceil_div(W, BLK), (unsigned)B), blk(BLK);
    const int h = (int)H, w = (int)W;
    // s1 (-> c, unused afterwards) and s2 (-> a)
    hipLaunchKernelGGL(hf_multiscale_kernel, grid, blk, 0, st, depth, c, a, h, w, max_depth, cam_scale, scale_2_80m);
    // s3 = close 5x5: a -> b (dilate) -> a (erode)
    hipLaunchKernelGGL((hf_morph_kernel<false>), grid, blk, 0, st, a, b, (const int*)nullptr, h, w, 5, 0);
    hipLaunchKernelGGL((hf_morph_kernel<true>), grid, blk, 0, st, b, a, (const int*)nullptr, h, w, 5, 0);
    // s4 = median where valid: a -> b
    hipLaunchKernelGGL(hf_median_kernel, grid, blk, 0, st, a, b, (const int*)nullptr, h, w);
    // s5 = 9x9 dilation into the empty pixels below the top row: b -> a
    hipLaunchKernelGGL(hf_top_row_kernel, cgrid, blk, 0, st, b, top, h, w);
    hipLaunchKernelGGL((hf_morph_kernel<false>), grid, blk, 0, st, b, a, top, h, w, 9, 1);
    // s7: six masked 5x5 dilations, top row from s5: a -> b -> a ...  (ends in a)
    hipLaunchKernelGGL(hf_top_row_kernel, cgrid, blk, 0, st, a, top, h, w);
    float* cur = a;
    float* nxt = b;
    for (int i = 0; i < 6; ++i) {
        hipLaunchKernelGGL((hf_morph_kernel<false>), grid, blk, 0, st, cur, nxt, top, h, w, 5, 2);
        float* t = cur; cur = nxt; nxt = t;
    }
    // median where valid & below the top row: cur -> nxt; bilateral gated by the pre-median image (cur) -> out, inverted
    hipLaunchKernelGGL(hf_median_kernel, grid, blk, 0, st, cur, nxt, top, h, w);
    hipLaunchKernelGGL(hf_bilateral_invert_kernel, grid, blk, 0, st, nxt, cur, top, out, h, w, 0.5f, 2.0f, max_depth,
                       (float)cam_scale, (float)scale_2_80m);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
