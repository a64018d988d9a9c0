// ffb6d_amd/csrc/inputs.hip -- on-device input pipeline in front of the index pyramid:
// valid-pixel sampling and point assembly of the reference's Dataset.get_item
//   ffb6d/datasets/linemod/linemod_dataset.py:262-289 (same block: datasets/ycb/ycb_dataset.py:217-245)
//       choose   = indices of pixels with depth > 1e-6
//       more than N valid: a uniformly random N-subset (shuffled 0/1 mask), fewer: np.pad(..., 'wrap')
//       then a uniformly random order (np.random.shuffle)
//       cld / rgb_pt / nrm_pt = per-point rows of the xyz, colour and normal images; cld_rgb_nrm = their concatenation
//
// Here, without any host round trip (the reference's nonzero / shuffle run in numpy inside DataLoader workers):
//   1. every pixel gets a 32-bit sort key  valid ? hash32(seed, frame, pixel) : 0xffffffff;
//   2. ONE stable segmented radix sort (csrc/seg_sort.hip: the frames are the segments; round 5 -- rocprim::radix_sort_pairs on 64-bit
//      (frame, hash) keys before: the same permutation, 0.68 -> see profiles/r05_inputs_bench.json) puts each frame's valid pixels
//      first, in uniformly random order;
//   3. the first N entries of a frame's segment are the sample: a uniformly random N-subset in uniformly random order.
//      A frame with fewer than N valid pixels repeats its permutation cyclically (the reference repeats the ascending
//      list and shuffles afterwards: same multiset up to which pixels get the extra copy, and the prefix of any length --
//      what the index pyramid's "random" sub-sampling takes, linemod_dataset.py:322-323 -- is a random subset either way);
//   4. one gather kernel writes choose, the point cloud [B,N,3] and cld_rgb_nrm [B,9,N].
// Distribution-equivalent to the reference, not bit-identical (the reference draws from numpy's global generator).
#include "common.h"
#include "ffb6d_ops.h"

#include "seg_sort.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

__device__ __forceinline__ uint32_t mix32(uint32_t x)     // murmur3 finaliser: bijective 32-bit mixing
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// A workgroup keys 2048 consecutive pixels of a frame and adds its count of valid ones to n_valid[b] with ONE atomic (round 5: one
// atomic per wave on eight addresses was 38 400 serialised device-scope atomics per batch -- 437 us of a 24-us kernel,
// profiles/r05_inputs_bench.json).
constexpr int KEYS_PER_BLOCK = 2048;
__global__ void __launch_bounds__(BLK)
sample_keys_kernel(const float* __restrict__ depth, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                   int32_t* __restrict__ n_valid, int HW, float min_depth, uint32_t seed_lo, uint32_t seed_hi)
{
    __shared__ int wave_count[BLK / 64];
    const int b = blockIdx.y;
    int mine = 0;
#pragma unroll
    for (int r = 0; r < KEYS_PER_BLOCK / BLK; ++r) {
        const int pix = blockIdx.x * KEYS_PER_BLOCK + r * BLK + threadIdx.x;
        bool valid = false;
        if (pix < HW) {
            valid = depth[(size_t)b * HW + pix] > min_depth;           // NaN compares false: invalid
            const uint32_t h = mix32(mix32((uint32_t)pix ^ seed_lo) + 0x9e3779b9u * (uint32_t)(b + 1) + seed_hi);
            // valid keys stay below 0xffffffff so that every invalid pixel sorts behind every valid one
            keys[(size_t)b * HW + pix] = valid ? (h == 0xffffffffu ? 0xfffffffeu : h) : 0xffffffffu;
            vals[(size_t)b * HW + pix] = (uint32_t)pix;
        }
        mine += (int)__popcll(__ballot(valid));                        // (the wave's count: the same number in every lane)
    }
    if ((threadIdx.x & 63) == 0) wave_count[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) total += wave_count[w];
        if (total) atomicAdd(n_valid + b, total);
    }
}

// choose[b,j] = sorted pixel (j mod n_valid[b]); cld [B,N,3] and cld_rgb_nrm [B,9,N] gathered from the images
template <typename RgbT>
__global__ void __launch_bounds__(BLK)
assemble_points_kernel(const uint32_t* __restrict__ sorted, const int32_t* __restrict__ n_valid, const float* __restrict__ xyz,
                       const RgbT* __restrict__ rgb, const float* __restrict__ nrm, long long* __restrict__ choose,
                       float* __restrict__ cld, float* __restrict__ crn, int HW, int N)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * BLK + threadIdx.x;
    if (j >= N) return;
    const int nv = n_valid[b];
    const uint32_t pix = nv > 0 ? sorted[(size_t)b * HW + (j % nv)] : 0u;
    choose[(size_t)b * N + j] = (long long)pix;
    if (!xyz) return;                         // indices only
    const float* x = xyz + (size_t)b * 3 * HW + pix;
    const RgbT* c = rgb + (size_t)b * 3 * HW + pix;
    const float* n = nrm + (size_t)b * 3 * HW + pix;
    float* o = crn + (size_t)b * 9 * N + j;
    float* p = cld + ((size_t)b * N + j) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = x[(size_t)a * HW];
        p[a] = v;
        o[(size_t)a * N] = v;
        o[(size_t)(3 + a) * N] = (float)c[(size_t)a * HW];
        o[(size_t)(6 + a) * N] = n[(size_t)a * HW];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Surface normals from a depth image.  The reference calls the third-party package
// `normalSpeed` (github.com/hfutcgncas/normalSpeed, installed from source per the reference's README.md:66-72, no version
// pin; NOT vendored in the reference tree and absent from this image):
//     nrm_map = normalSpeed.depth_normal(dpt_mm, K[0][0], K[1][1], 5, 2000, 20, False)
// at linemod_dataset.py:252-254, ycb_dataset.py:208-210 and utils/basic_utils.py:323.  Restated here from its published
// algorithm -- OpenCV's LINE-MOD depth normals (opencv/modules/rgbd, `accumBilateral` least squares): for a pixel of depth
// d < distance_threshold take the 8 neighbours at offset (+-r, +-r) / (+-r, 0) / (0, +-r), r = kernel_size; a neighbour
// whose depth differs from d by less than difference_threshold contributes (i, j, delta) to the normal equations
//     A = sum [i*i, i*j; i*j, j*j],  b = sum [i*delta, j*delta]            (integer arithmetic)
// and the normal is  (fx * (A11 b0 - A01 b1),  fy * (-A01 b0 + A00 b1),  -det(A) * d)  normalised; pixels in the r-wide border,
// pixels with d >= distance_threshold and degenerate systems stay (0, 0, 0).  PARITY UNPINNED: no reference output can be
// produced here; parity is anchored on the call sites' arguments and checked against the CPU restatement (oracle/inputs_ref.py).
// ---------------------------------------------------------------------------------------------------------------
template <typename DepthT>
__global__ void __launch_bounds__(BLK)
depth_normal_kernel(const DepthT* __restrict__ depth, float* __restrict__ out, int H, int W, double fx, double fy, int r,
                    int dist_thr, int diff_thr)
{
    const int b = blockIdx.y;
    const int pix = blockIdx.x * BLK + threadIdx.x;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix - y * W;
    const DepthT* img = depth + (size_t)b * H * W;
    auto at = [&](int yy, int xx) -> long long { return (long long)(unsigned short)(long long)img[(size_t)yy * W + xx]; };
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (y >= r && y < H - r - 1 && x >= r && x < W - r - 1) {
        const long long d = at(y, x);
        if (d < dist_thr) {
            long long a00 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int t = n < 4 ? n : n + 1;               // the 3 x 3 stencil without its centre
                const long long i = (t % 3 - 1) * r, j = (t / 3 - 1) * r;
                const long long delta = at(y + (int)j, x + (int)i) - d;
                if ((delta < 0 ? -delta : delta) < diff_thr) {
                    a00 += i * i; a01 += i * j; a11 += j * j;
                    b0 += i * delta; b1 += j * delta;
                }
            }
            const long long det = a00 * a11 - a01 * a01;
            const long long ddx = a11 * b0 - a01 * b1;
            const long long ddy = -a01 * b0 + a00 * b1;
            const float fxn = (float)(fx * (double)ddx), fyn = (float)(fy * (double)ddy), fzn = (float)(-det * d);
            const float len = sqrtf(fxn * fxn + fyn * fyn + fzn * fzn);
            if (len > 0.f) {
                const float inv = 1.0f / len;
                nx = fxn * inv; ny = fyn * inv; nz = fzn * inv;
            }
        }
    }
    float* o = out + (size_t)b * 3 * H * W + pix;
    o[0] = nx;
    o[(size_t)H * W] = ny;
    o[(size_t)2 * H * W] = nz;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct SampleLayout { size_t keys_in, keys_out, vals_in, vals_out, temp, temp_bytes, total; };

segsort::Plan sample_plan(int64_t B, int64_t HW)
{
    segsort::Plan p;
    p.ngroups = 1;
    p.B = (int)B;
    p.g[0].pos0 = 0;
    p.g[0].S = (int)HW;
    return p;
}

SampleLayout sample_layout(int64_t B, int64_t HW)
{
    SampleLayout L;
    const size_t n = (size_t)B * HW;
    size_t off = 0;
    L.keys_in = off; off += align256(n * 4);
    L.keys_out = off; off += align256(n * 4);
    L.vals_in = off; off += align256(n * 4);
    L.vals_out = off; off += align256(n * 4);
    L.temp = off; L.temp_bytes = align256(segsort::temp_bytes(sample_plan(B, HW)) + 256); off += L.temp_bytes;
    L.total = off;
    return L;
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

extern "C" size_t ffb6d_sample_points_workspace_bytes(int64_t B, int64_t H, int64_t W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return sample_layout(B, H * W).total;
}

extern "C" int ffb6d_sample_points_f32(const float* depth, float min_depth, const float* xyz, const void* rgb, int rgb_is_u8,
                                       const float* nrm, uint64_t seed, int64_t* choose, float* cld, float* cld_rgb_nrm,
                                       int32_t* n_valid, int64_t B, int64_t H, int64_t W, int64_t N, void* workspace,
                                       size_t workspace_bytes, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && N >= 1, "sample_points: bad shape");
    FFB6D_REQUIRE(H * W < (1LL << 31) && B < 65536 && N < (1LL << 31), "sample_points: size too large");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(depth && choose && n_valid, "sample_points: null pointer");
    const bool gather = xyz || rgb || nrm || cld || cld_rgb_nrm;
    FFB6D_REQUIRE(!gather || (xyz && rgb && nrm && cld && cld_rgb_nrm),
                  "sample_points: the image sources and point outputs come together (or all NULL for indices only)");
    const int64_t HW = H * W;
    const SampleLayout L = sample_layout(B, HW);
    if (!workspace || workspace_bytes < L.total)
        return set_error(FFB6D_ERR_WORKSPACE, "sample_points: workspace of %zu bytes required, got %zu", L.total,
                         workspace ? workspace_bytes : (size_t)0);
    char* ws = static_cast<char*>(workspace);
    auto* keys_in = reinterpret_cast<uint32_t*>(ws + L.keys_in);
    auto* keys_out = reinterpret_cast<uint32_t*>(ws + L.keys_out);
    auto* vals_in = reinterpret_cast<uint32_t*>(ws + L.vals_in);
    auto* vals_out = reinterpret_cast<uint32_t*>(ws + L.vals_out);
    hipStream_t st = as_stream(stream);
    FFB6D_HIP_TRY(hipMemsetAsync(n_valid, 0, (size_t)B * sizeof(int32_t), st));
    hipLaunchKernelGGL(sample_keys_kernel, dim3((unsigned)ceil_div(HW, KEYS_PER_BLOCK), (unsigned)B), dim3(BLK), 0, st, depth, keys_in, vals_in,
                       n_valid, (int)HW, min_depth, (uint32_t)seed, (uint32_t)(seed >> 32));
    segsort::Plan plan = sample_plan(B, HW);
    bool in_alt = false;
    const hipError_t se = segsort::sort_pairs(plan, keys_in, vals_in, keys_out, vals_out, 32, ws + L.temp, L.temp_bytes, st, &in_alt);
    if (se != hipSuccess) return set_error(FFB6D_ERR_HIP, "sample_points: segmented sort failed: %s", hipGetErrorString(se));
    const uint32_t* sorted = in_alt ? vals_out : vals_in;
    const dim3 grid((unsigned)ceil_div(N, BLK), (unsigned)B);
    if (rgb_is_u8)
        hipLaunchKernelGGL((assemble_points_kernel<uint8_t>), grid, dim3(BLK), 0, st, sorted, n_valid, xyz,
                           static_cast<const uint8_t*>(rgb), nrm, reinterpret_cast<long long*>(choose), cld, cld_rgb_nrm, (int)HW, (int)N);
    else
        hipLaunchKernelGGL((assemble_points_kernel<float>), grid, dim3(BLK), 0, st, sorted, n_valid, xyz,
                           static_cast<const float*>(rgb), nrm, reinterpret_cast<long long*>(choose), cld, cld_rgb_nrm, (int)HW, (int)N);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_depth_normal(const void* depth_mm, int depth_is_u16, double fx, double fy, int kernel_size,
                                  int distance_threshold, int difference_threshold, int point_into_surface, float* out,
                                  int64_t B, int64_t H, int64_t W, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && H * W < (1LL << 31) && B < 65536, "depth_normal: bad shape");
    FFB6D_REQUIRE(kernel_size >= 1 && kernel_size < 64, "depth_normal: kernel_size must be in [1, 63]");
    FFB6D_REQUIRE(point_into_surface == 0, "depth_normal: point_into_surface = True has no call site in the reference and is not provided");
    if (B == 0) return FFB6D_OK;
    FFB6D_REQUIRE(depth_mm && out, "depth_normal: null pointer");
    const dim3 grid((unsigned)ceil_div(H * W, BLK), (unsigned)B);
    hipStream_t st = as_stream(stream);
    if (depth_is_u16)
        hipLaunchKernelGGL((depth_normal_kernel<uint16_t>), grid, dim3(BLK), 0, st, static_cast<const uint16_t*>(depth_mm), out, (int)H,
                           (int)W, fx, fy, kernel_size, distance_threshold, difference_threshold);
    else
        hipLaunchKernelGGL((depth_normal_kernel<float>), grid, dim3(BLK), 0, st, static_cast<const float*>(depth_mm), out, (int)H, (int)W,
                           fx, fy, kernel_size, distance_threshold, difference_threshold);
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Point sets of the index pyramid (linemod_dataset.py:299-323) in ONE launch: the cloud as [B,N,3] rows, its 16-byte coordinate table,
// the prefixes that are the coarser cloud levels, and the xyz image at the strides the pixel <-> point searches use.  These were 13
// ATen launches (slices, transposes, a pad) in front of the set preparation: 0.2 ms of dependent launches on the stream every index of
// the step waits for (profiles/r05_step_timeline.txt).  Pure copies: bit-identical to the slices.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace ffb6d {
namespace {

constexpr int PYR_MAX_LEVELS = 6, PYR_MAX_GRIDS = 4;

struct PyrSets {
    const float* cld; long long cld_bs, cld_cs, cld_ps;          // cloud source: element strides of frame, coordinate, point
    float* level_out[PYR_MAX_LEVELS]; int level_n[PYR_MAX_LEVELS];
    float* table;
    const float* dpt;                                             // [B,3,H,W]
    float* grid_out[PYR_MAX_GRIDS]; int grid_m[PYR_MAX_GRIDS], grid_h[PYR_MAX_GRIDS], grid_w[PYR_MAX_GRIDS];
    int B, N, n_levels, H, W, n_grids, s0, h0, w0;
    unsigned cloud_blocks;
};

__global__ void __launch_bounds__(BLK)
pyramid_sets_kernel(const PyrSets p)
{
    if (blockIdx.x < p.cloud_blocks) {
        const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
        if (t >= (long long)p.B * p.N) return;
        const int b = (int)(t / p.N), n = (int)(t - (long long)b * p.N);
        const float* s = p.cld + b * p.cld_bs + n * p.cld_ps;
        const float x = s[0], y = s[p.cld_cs], z = s[2 * p.cld_cs];
        if (p.table) reinterpret_cast<float4*>(p.table)[t] = make_float4(x, y, z, 0.f);
#pragma unroll
        for (int k = 0; k < PYR_MAX_LEVELS; ++k)
            if (k < p.n_levels && n < p.level_n[k]) {
                float* o = p.level_out[k] + ((long long)b * p.level_n[k] + n) * 3;
                o[0] = x; o[1] = y; o[2] = z;
            }
        return;
    }
    // one thread per pixel of the finest grid (stride s0); the coarser grids (stride s0 * m) are subsets of it
    const long long t = (long long)(blockIdx.x - p.cloud_blocks) * BLK + threadIdx.x;
    const long long per = (long long)p.h0 * p.w0;
    if (t >= p.B * per) return;
    const int b = (int)(t / per);
    const int r = (int)(t - b * per);
    const int gy = r / p.w0, gx = r - gy * p.w0;
    const size_t hw = (size_t)p.H * p.W;
    const float* s = p.dpt + (size_t)b * 3 * hw + (size_t)(gy * p.s0) * p.W + (size_t)gx * p.s0;
    const float x = s[0], y = s[hw], z = s[2 * hw];
#pragma unroll
    for (int k = 0; k < PYR_MAX_GRIDS; ++k)
        if (k < p.n_grids) {
            const int m = p.grid_m[k];
            const int Y = gy / m, X = gx / m;
            if (Y * m == gy && X * m == gx && Y < p.grid_h[k] && X < p.grid_w[k]) {
                float* o = p.grid_out[k] + ((size_t)b * p.grid_h[k] * p.grid_w[k] + (size_t)Y * p.grid_w[k] + X) * 3;
                o[0] = x; o[1] = y; o[2] = z;
            }
        }
}

}  // namespace
}  // namespace ffb6d

extern "C" int ffb6d_pyramid_sets_f32(const float* cloud, int64_t cloud_frame_stride, int64_t cloud_coord_stride,
                                      int64_t cloud_point_stride, int64_t B, int64_t N, int n_levels, const int64_t* level_n,
                                      float* const* level_out, float* table, const float* dpt_xyz, int64_t H, int64_t W,
                                      int n_grids, const int* strides, float* const* grid_out, ffb6d_stream_t stream)
{
    using namespace ffb6d;
    FFB6D_REQUIRE(B >= 0 && N >= 0 && n_levels >= 0 && n_levels <= PYR_MAX_LEVELS && n_grids >= 0 && n_grids <= PYR_MAX_GRIDS,
                  "pyramid_sets: at most %d cloud levels and %d grids", PYR_MAX_LEVELS, PYR_MAX_GRIDS);
    FFB6D_REQUIRE(B * N < (1LL << 31) && H >= 0 && W >= 0 && B * 3 * H * W < (1LL << 40), "pyramid_sets: too large");
    PyrSets p = {};
    const bool cloud_part = (n_levels > 0 || table) && B * N > 0;
    if (cloud_part) {
        FFB6D_REQUIRE(cloud && (n_levels == 0 || (level_n && level_out)), "pyramid_sets: null cloud pointer");
        FFB6D_REQUIRE(!table || (reinterpret_cast<uintptr_t>(table) & 15) == 0, "pyramid_sets: the coordinate table must be 16-byte aligned");
        p.cld = cloud; p.cld_bs = cloud_frame_stride; p.cld_cs = cloud_coord_stride; p.cld_ps = cloud_point_stride;
        p.table = table; p.n_levels = n_levels;
        for (int k = 0; k < n_levels; ++k) {
            FFB6D_REQUIRE(level_n[k] >= 0 && level_n[k] <= N && (level_out[k] || level_n[k] == 0), "pyramid_sets: level %d: bad size or null output", k);
            p.level_n[k] = (int)level_n[k]; p.level_out[k] = level_out[k];
        }
        p.cloud_blocks = (unsigned)ceil_div(B * N, (int64_t)BLK);
    }
    p.B = (int)B; p.N = (int)N;
    long long grid_threads = 0;
    if (n_grids > 0 && B > 0) {
        FFB6D_REQUIRE(dpt_xyz && strides && grid_out && H >= 1 && W >= 1, "pyramid_sets: null image pointer");
        int s0 = strides[0];
        for (int k = 0; k < n_grids; ++k) {
            FFB6D_REQUIRE(strides[k] >= 1 && strides[k] <= H && strides[k] <= W && grid_out[k], "pyramid_sets: grid %d: bad stride or null output", k);
            s0 = strides[k] < s0 ? strides[k] : s0;
        }
        p.dpt = dpt_xyz; p.H = (int)H; p.W = (int)W; p.n_grids = n_grids; p.s0 = s0; p.h0 = (int)(H / s0); p.w0 = (int)(W / s0);
        for (int k = 0; k < n_grids; ++k) {
            FFB6D_REQUIRE(strides[k] % s0 == 0, "pyramid_sets: every stride must be a multiple of the smallest one");
            p.grid_m[k] = strides[k] / s0; p.grid_h[k] = (int)(H / strides[k]); p.grid_w[k] = (int)(W / strides[k]);
            p.grid_out[k] = grid_out[k];
        }
        grid_threads = (long long)B * p.h0 * p.w0;
    }
    const long long blocks = (long long)p.cloud_blocks + ceil_div((int64_t)grid_threads, (int64_t)BLK);
    if (blocks == 0) return FFB6D_OK;
    FFB6D_REQUIRE(blocks < (1LL << 31), "pyramid_sets: too many workgroups");
    hipLaunchKernelGGL(pyramid_sets_kernel, dim3((unsigned)blocks), dim3(BLK), 0, as_stream(stream), p);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
