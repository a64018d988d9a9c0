// ffb6d_amd/csrc/knn_pruned.hip -- exact KNN with spatial pruning for gfx950 (MI355X).
//
// Same contract as knn.hip (reference: knn_.cxx:104-135, nanoflann.hpp:79-145,323-348: exact
// K nearest, ascending f32 squared distance ((dx*dx+dy*dy)+dz*dz), lowest index on ties), but
// the work is cut from S*Q point pairs to a few hundred per query:
//
//   prepare   a point set is put into Morton (Z-curve) order once: per-frame bounding box ->
//             30-bit Morton keys -> one device radix sort of (frame<<32 | key) -> gather into
//             float4 {x, y, z, original index} + one axis-aligned box per 64-point tile;
//   search    one lane owns one query (queries are themselves Morton ordered, so the 64 lanes
//             of a wave sit next to each other in space).  The wave binary-searches the tile
//             where its queries would fall and walks the tiles outwards from there; a tile is
//             entered only if its box is within the current K-th distance of some lane
//             (wave-uniform branch on __any), so after the first couple of tiles almost
//             everything is skipped.  Tile boxes and tile points are wave-uniform, i.e. they
//             come through the scalar cache into SGPRs -- no LDS staging, no bank conflicts;
//             the per-lane top-K lives in registers and takes candidates through the same
//             per-lane LDS queue + wave-wide drain as the brute-force kernel.
//
// Exactness does not depend on the ordering heuristics: the box lower bound is evaluated with
// the reference's operation order ((gx*gx+gy*gy)+gz*gz) on monotone-rounded gaps, hence it is
// <= the rounded distance of every point in the box; tiles with bound <= current K-th distance
// are always entered; candidates are compared as (distance, original index) pairs, so the
// visiting order cannot change which neighbour wins a tie.
#include "common.h"

#include <rocprim/rocprim.hpp>      // 64-bit sort keys only (more than 256 (set, frame) segments in one call)

#include "seg_sort.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <type_traits>

namespace ffb6d {
namespace {

// Optional instrumentation (bench.py): when set, the search kernels add the number of point pairs whose distance they
// actually evaluated to *g_pair_counter (one atomic per wave).  SURVEY.md section 8d asks for KNN as evaluated pairs/s
// against the fp32 VALU roof; the pruned search evaluates a small, data-dependent fraction of the S x Q candidates.
__device__ unsigned long long* g_pair_counter = nullptr;

__device__ __forceinline__ void publish_pairs(unsigned int mine)
{
    unsigned long long* ctr = g_pair_counter;
    if (!ctr) return;
    unsigned long long v = mine;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(ctr, v);
}

constexpr int PT = 64;      // points per tile (one box per tile)
constexpr int BLK = 256;
constexpr int QCAP = 16;    // per-lane queue slots
constexpr int GRP = 8;      // candidates per unrolled group

constexpr int CELL_BITS = 15;                 // coarse Z-curve cells (top bits of the 30-bit key)
constexpr int CELL_SHIFT = 30 - CELL_BITS;
// The sort that brings a set into Morton order sees only the top SORT_BITS of the 30-bit key (8 bits per axis: with at most 76 800 points
// in 16 M cells nearly every point still has a cell of its own) under the (set, frame) segment number: 32-bit sort keys for up to 256
// segments -- two thirds of the bytes a 64-bit key + 32-bit value sort moves (the sort was 0.45 ms of the step, profiles/r04_rocprofv3_
// kernel_stats_steady_state.txt).  The order INSIDE a cell is the input order (stable sort); no search result depends on it.
constexpr int SORT_BITS = 24;
constexpr int SORT_DROP = 30 - SORT_BITS;
constexpr int NCELL = 1 << CELL_BITS;

constexpr int FAN = 16;                       // tiles per level-2 box

struct Layout {
    int64_t S_pad, nt, nt2;
    size_t pts_off, box_off, box2_off, key_off, frame_off, cell_off, total;
};

Layout layout(int64_t B, int64_t S)
{
    Layout l;
    l.S_pad = ceil_div(S, PT) * PT;
    l.nt = l.S_pad / PT;
    l.nt2 = ceil_div(l.nt, FAN);
    size_t o = 0;
    l.pts_off = o;   o += (size_t)B * l.S_pad * sizeof(float4);
    l.box_off = o;   o += (size_t)B * l.nt * 2 * sizeof(float4);
    l.box2_off = o;  o += (size_t)B * l.nt2 * 2 * sizeof(float4);
    l.key_off = o;   o += (size_t)B * l.S_pad * sizeof(uint32_t);
    l.frame_off = o; o += (size_t)B * 8 * sizeof(float);
    l.cell_off = o;  o += (size_t)B * NCELL * sizeof(uint32_t);   // cell -> tile holding its first point
    l.total = (o + 255) / 256 * 256;
    return l;
}

// ---------------------------------------------------------------- prepare -------------
__device__ __forceinline__ int f2ord(float f)
{
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);   // monotone float -> int
}
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }

// per-frame bounding box -> quantisation frame {lo.xyz, 0, scale.xyz, 0}; one 1024-thread
// block per frame (a frame is at most a few hundred thousand points)
__device__ __forceinline__ void frame_body(const float* __restrict__ pts, int S, float* __restrict__ frame, int b)
{
    __shared__ float red[6][16];
    const float* p = pts + (size_t)b * S * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    // four points per thread and trip, their twelve loads issued together: one block walks a whole frame (76 800 points for the
    // stride-2 image grid) and a trip is one memory round trip -- 27.6 us per call with one point per trip (profiles/r05_knn_pmc.txt)
    for (int i0 = threadIdx.x; i0 < S; i0 += 4 * 1024) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 1024, S - 1);         // past the end: the last point again
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = p[(size_t)i * 3 + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lo[c] = fminf(lo[c], v[u][c]);
                hi[c] = fmaxf(hi[c], v[u][c]);
            }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[c][w] = lo[c]; red[3 + c][w] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        float l = INFINITY, h = -INFINITY;
        for (int i = 0; i < 16; ++i) { l = fminf(l, red[c][i]); h = fmaxf(h, red[3 + c][i]); }
        const float ext = h - l;
        frame[b * 8 + c] = l;
        frame[b * 8 + 4 + c] = (ext > 0.f && ext < INFINITY) ? 1024.f / ext : 0.f;
    }
    if (threadIdx.x == 3) { frame[b * 8 + 3] = 0.f; frame[b * 8 + 7] = 0.f; }
}

__global__ void __launch_bounds__(1024)
frame_kernel(const float* __restrict__ pts, int S, float* __restrict__ frame)
{
    frame_body(pts, S, frame, blockIdx.x);
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// frame = {lo.x, lo.y, lo.z, 0, scale.x, scale.y, scale.z, 0}
__device__ __forceinline__ uint32_t morton_key(float x, float y, float z, const float* __restrict__ fr)
{
    const float cx = fminf(fmaxf((x - fr[0]) * fr[4], 0.f), 1023.f);
    const float cy = fminf(fmaxf((y - fr[1]) * fr[5], 0.f), 1023.f);
    const float cz = fminf(fmaxf((z - fr[2]) * fr[6], 0.f), 1023.f);
    return spread10((uint32_t)cx) | (spread10((uint32_t)cy) << 1) | (spread10((uint32_t)cz) << 2);
}

// `seg` = the sort segment of frame b: keys of one segment stay together and in segment order
template <typename KeyT>
__device__ __forceinline__ void morton_body(const float* __restrict__ pts, int S, const float* __restrict__ frame,
                                            KeyT* __restrict__ keys, uint32_t* __restrict__ vals, int b, unsigned seg)
{
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= S) return;
    const float* p = pts + ((size_t)b * S + i) * 3;
    const uint32_t k = morton_key(p[0], p[1], p[2], frame + b * 8);
    keys[(size_t)b * S + i] = ((KeyT)seg << SORT_BITS) | (KeyT)(k >> SORT_DROP);
    vals[(size_t)b * S + i] = (uint32_t)i;
}

template <typename KeyT>
__global__ void __launch_bounds__(BLK)
morton_kernel(const float* __restrict__ pts, int S, const float* __restrict__ frame, KeyT* __restrict__ keys, uint32_t* __restrict__ vals)
{
    morton_body(pts, S, frame, keys, vals, blockIdx.y, blockIdx.y);
}

// one wave per tile: gather the sorted points, build the tile box
template <typename KeyT>
__device__ __forceinline__ void gather_box_body(const float* __restrict__ pts, int S, int S_pad, int nt,
                                                const KeyT* __restrict__ skeys, const uint32_t* __restrict__ perm,
                                                float4* __restrict__ out_pts, float4* __restrict__ boxes,
                                                uint32_t* __restrict__ out_keys, int b)
{
    const int t = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (t >= nt) return;
    const int lane = threadIdx.x & 63;
    const int slot = t * PT + lane;
    float4 p = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xffffffffu));
    uint32_t key = 0xffffffffu;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (slot < S) {
        const uint32_t src = perm[(size_t)b * S + slot];
        const float* s = pts + ((size_t)b * S + src) * 3;
        p = make_float4(s[0], s[1], s[2], __uint_as_float(src));
        key = ((uint32_t)skeys[(size_t)b * S + slot] & ((1u << SORT_BITS) - 1u)) << SORT_DROP;       // the sorted bits of the 30-bit key
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64));
        }
    }
    out_pts[(size_t)b * S_pad + slot] = p;
    out_keys[(size_t)b * S_pad + slot] = key;
    if (lane == 0) {
        boxes[((size_t)b * nt + t) * 2 + 0] = make_float4(lo[0], lo[1], lo[2], 0.f);
        boxes[((size_t)b * nt + t) * 2 + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

template <typename KeyT>
__global__ void __launch_bounds__(BLK)
gather_box_kernel(const float* __restrict__ pts, int S, int S_pad, int nt,
                  const KeyT* __restrict__ skeys, const uint32_t* __restrict__ perm,
                  float4* __restrict__ out_pts, float4* __restrict__ boxes, uint32_t* __restrict__ out_keys)
{
    gather_box_body(pts, S, S_pad, nt, skeys, perm, out_pts, boxes, out_keys, blockIdx.y);
}

// level-2 boxes: the hull of FAN consecutive tile boxes
__device__ __forceinline__ void box2_body(const float4* __restrict__ boxes, int nt, int nt2, float4* __restrict__ boxes2, int b)
{
    const int g = blockIdx.x * BLK + threadIdx.x;
    if (g >= nt2) return;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
    for (int t = g * FAN; t < min(nt, (g + 1) * FAN); ++t) {
        const float4 a = boxes[((size_t)b * nt + t) * 2], c = boxes[((size_t)b * nt + t) * 2 + 1];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
        hi.x = fmaxf(hi.x, c.x); hi.y = fmaxf(hi.y, c.y); hi.z = fmaxf(hi.z, c.z);
    }
    boxes2[((size_t)b * nt2 + g) * 2] = lo;
    boxes2[((size_t)b * nt2 + g) * 2 + 1] = hi;
}

__global__ void __launch_bounds__(BLK)
box2_kernel(const float4* __restrict__ boxes, int nt, int nt2, float4* __restrict__ boxes2)
{
    box2_body(boxes, nt, nt2, boxes2, blockIdx.y);
}

// cell -> index of the tile that holds the first point whose key is >= (cell << CELL_SHIFT)
__device__ __forceinline__ void cell_table_body(const uint32_t* __restrict__ keys, int S, int S_pad, int nt,
                                                uint32_t* __restrict__ table, int b)
{
    const int c = blockIdx.x * BLK + threadIdx.x;
    if (c >= NCELL) return;
    const uint32_t* k = keys + (size_t)b * S_pad;
    const uint32_t want = (uint32_t)c << CELL_SHIFT;
    int lo = 0, hi = S;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (k[mid] < want) lo = mid + 1; else hi = mid;
    }
    table[(size_t)b * NCELL + c] = (uint32_t)min(lo / PT, nt - 1);
}

__global__ void __launch_bounds__(BLK)
cell_table_kernel(const uint32_t* __restrict__ keys, int S, int S_pad, int nt, uint32_t* __restrict__ table)
{
    cell_table_body(keys, S, S_pad, nt, table, blockIdx.y);
}

// ---- several point sets in one go (the 22 searches of an index pyramid touch a handful of sets, all known up front:
// linemod_dataset.py:299-353): blockIdx.z = set, the kernels above with one more grid dimension, and ONE radix sort over the
// concatenation of all sets with the sort segment (set, frame) in the key's high bits -- a handful of launches instead of
// a dozen per set, and the sort's fixed costs are paid once.
constexpr int MAX_SETS = 8;
struct SetDesc {
    const float* pts;
    float4 *out_pts, *boxes, *boxes2;
    uint32_t *out_keys, *cell;
    float* frame;
    int S, S_pad, nt, nt2;
    long long pos0;       // first element of the set in the concatenated key / value arrays
};
struct MultiDesc {
    SetDesc s[MAX_SETS];
    int B;
};

__global__ void __launch_bounds__(1024)
frame_multi_kernel(const MultiDesc m)
{
    const SetDesc& d = m.s[blockIdx.y];
    frame_body(d.pts, d.S, d.frame, blockIdx.x);
}

template <typename KeyT>
__global__ void __launch_bounds__(BLK)
morton_multi_kernel(const MultiDesc m, KeyT* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const SetDesc& d = m.s[blockIdx.z];
    morton_body(d.pts, d.S, d.frame, keys + d.pos0, vals + d.pos0, blockIdx.y, blockIdx.z * m.B + blockIdx.y);
}

template <typename KeyT>
__global__ void __launch_bounds__(BLK)
gather_box_multi_kernel(const MultiDesc m, const KeyT* __restrict__ skeys, const uint32_t* __restrict__ perm)
{
    const SetDesc& d = m.s[blockIdx.z];
    gather_box_body(d.pts, d.S, d.S_pad, d.nt, skeys + d.pos0, perm + d.pos0, d.out_pts, d.boxes, d.out_keys, blockIdx.y);
}

__global__ void __launch_bounds__(BLK)
box2_multi_kernel(const MultiDesc m)
{
    const SetDesc& d = m.s[blockIdx.z];
    box2_body(d.boxes, d.nt, d.nt2, d.boxes2, blockIdx.y);
}

__global__ void __launch_bounds__(BLK)
cell_table_multi_kernel(const MultiDesc m)
{
    const SetDesc& d = m.s[blockIdx.z];
    cell_table_body(d.out_keys, d.S, d.S_pad, d.nt, d.cell, blockIdx.y);
}

// ---------------------------------------------------------------- search --------------
// running top-K as packed 64-bit keys (distance bits << 32 | original index): non-negative
// floats order like their bit patterns, so one unsigned 64-bit compare is the lexicographic
// (distance, index) compare the tie rule needs
typedef unsigned long long u64;
__device__ __forceinline__ u64 pack_key(float d, uint32_t i) { return ((u64)__float_as_uint(d) << 32) | i; }
constexpr u64 KEY_EMPTY = ((u64)0x7f7fffffu << 32) | 0xffffffffu;   // (FLT_MAX, 0xffffffff)

template <int K>
struct TopKL {
    u64 k[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < K; ++j) k[j] = KEY_EMPTY;
    }
    __device__ __forceinline__ float worst() const { return __uint_as_float((uint32_t)(k[K - 1] >> 32)); }
    __device__ __forceinline__ bool beats_last(u64 nk) const { return nk < k[K - 1]; }
    // precondition: beats_last(nk).  One compare per slot: slot j takes its left neighbour when
    // that neighbour sorts after nk, takes nk when only the slot itself does.
    __device__ __forceinline__ void insert(u64 nk)
    {
        bool cur_after = true;
#pragma unroll
        for (int j = K - 1; j >= 1; --j) {
            const bool prev_after = k[j - 1] > nk;
            k[j] = prev_after ? k[j - 1] : (cur_after ? nk : k[j]);
            cur_after = prev_after;
        }
        if (cur_after) k[0] = nk;
    }
};

__device__ __forceinline__ float sqdist3(float qx, float qy, float qz, float px, float py, float pz)
{
    const float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// lower bound of sqdist3(q, p) over all p inside [lo, hi]; same op order, monotone rounding
__device__ __forceinline__ float box_bound(float qx, float qy, float qz, const float4& lo, const float4& hi)
{
    const float gx = fmaxf(fmaxf(__fsub_rn(lo.x, qx), __fsub_rn(qx, hi.x)), 0.f);
    const float gy = fmaxf(fmaxf(__fsub_rn(lo.y, qy), __fsub_rn(qy, hi.y)), 0.f);
    const float gz = fmaxf(fmaxf(__fsub_rn(lo.z, qz), __fsub_rn(qz, hi.z)), 0.f);
    return __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
}

// conservative bound between the wave's query box [qlo,qhi] and a tile box [lo,hi]
__device__ __forceinline__ float boxbox_bound(const float qlo[3], const float qhi[3], const float4& lo, const float4& hi)
{
    const float gx = fmaxf(fmaxf(__fsub_rn(lo.x, qhi[0]), __fsub_rn(qlo[0], hi.x)), 0.f);
    const float gy = fmaxf(fmaxf(__fsub_rn(lo.y, qhi[1]), __fsub_rn(qlo[1], hi.y)), 0.f);
    const float gz = fmaxf(fmaxf(__fsub_rn(lo.z, qhi[2]), __fsub_rn(qlo[2], hi.z)), 0.f);
    return __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
}

// value of `v` in lane `l` (l wave-uniform), bit-exact
__device__ __forceinline__ float lane_bcast(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// grid = (ceil(Q_pad / 256), B); queries come Morton ordered with their original index in .w.
// Each wave works on its own: 64 neighbouring queries, tiles tested 64 at a time (one tile box
// per lane against the wave's query box), surviving tiles fetched with one coalesced load and
// broadcast through a wave-private LDS slice.
template <int K>
__device__ __forceinline__ void
knn_pruned_body(const float4* __restrict__ spts, const float4* __restrict__ boxes,
                const uint32_t* __restrict__ skeys, const float* __restrict__ sframe,
                const uint32_t* __restrict__ scell, int S, int S_pad, int nt,
                const float4* __restrict__ qpts, int Q, int Q_pad,
                int64_t* __restrict__ idx64, int32_t* __restrict__ idx32, float* __restrict__ dist,
                int Kout, const int blk_x, const int b)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile_all = reinterpret_cast<float4*>(smem);                               // [4 waves][PT]
    uint2* queue = reinterpret_cast<uint2*>(smem + (BLK / 64) * PT * sizeof(float4));   // [QCAP][BLK]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    float4* tile = tile_all + (tid >> 6) * PT;
    const int slot = blk_x * BLK + tid;
    const float4 q = qpts[(size_t)b * Q_pad + min(slot, Q_pad - 1)];
    const uint32_t q_orig = __float_as_uint(q.w);
    const bool live = slot < Q_pad && q_orig != 0xffffffffu;
    const unsigned long long alive = __ballot(live);
    if (!alive) return;   // whole wave is padding (wave-uniform exit; no block barriers below)

    const float4* sp = spts + (size_t)b * S_pad;
    const float4* bx = boxes + (size_t)b * nt * 2;

    // the wave's query box
    float qlo[3], qhi[3];
    qlo[0] = wave_min(live ? q.x : INFINITY);  qhi[0] = wave_max(live ? q.x : -INFINITY);
    qlo[1] = wave_min(live ? q.y : INFINITY);  qhi[1] = wave_max(live ? q.y : -INFINITY);
    qlo[2] = wave_min(live ? q.z : INFINITY);  qhi[2] = wave_max(live ? q.z : -INFINITY);

    // every lane looks up the tile its own query falls into on the support's Z-curve
    const uint32_t key = morton_key(q.x, q.y, q.z, sframe + b * 8);
    const int my_tile = live ? (int)scell[(size_t)b * NCELL + (key >> CELL_SHIFT)] : 0;
    const int mid_lane = ((alive >> 32) & 1ull) ? 32 : (int)(__ffsll((long long)alive) - 1);
    const int t0 = __builtin_amdgcn_readlane(my_tile, mid_lane);   // where the sweep starts

    TopKL<K> top;
    top.init();
    float worst = live ? FLT_MAX : -1.0f;   // dead lanes never ask for a tile or a candidate
    int cnt = 0;

    auto drain = [&]() {
        if constexpr (K > 1) {
            for (int it = 0; __any(it < cnt); ++it) {
                if (it < cnt) {
                    const uint2 e = queue[it * BLK + tid];
                    const u64 nk = ((u64)e.x << 32) | e.y;
                    if (top.beats_last(nk)) top.insert(nk);
                }
            }
            cnt = 0;
            worst = top.worst();
        }
    };

    // take one candidate (d, sidx) for this lane if it can still matter
    auto offer = [&](float d, uint32_t sidx, float limit) {
        if constexpr (K == 1) {
            const u64 nk = pack_key(d, sidx);
            if (d <= limit && top.beats_last(nk)) { top.k[0] = nk; worst = d; }
        } else {
            if (d <= limit) {
                queue[cnt * BLK + tid] = make_uint2(__float_as_uint(d), sidx);
                cnt++;
            }
        }
    };

    // 1) seed: each lane scans the tile of its OWN query (per-lane addresses), which gives it a
    //    tight K-th distance before the wave-uniform sweep starts; otherwise a lane would meet
    //    ever closer tiles during the sweep and insert nearly every point of them.
    unsigned int pairs = live ? PT : 0;      // evaluated distances of this lane (seed tile below)
    {
        const float4* tp = sp + (size_t)my_tile * PT;
#pragma unroll 1
        for (int j0 = 0; j0 < PT; j0 += GRP) {
            float4 p[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) p[u] = tp[j0 + u];
#pragma unroll
            for (int u = 0; u < GRP; ++u)
                offer(sqdist3(q.x, q.y, q.z, p[u].x, p[u].y, p[u].z), __float_as_uint(p[u].w), worst);
            if constexpr (K > 1) {
                if (__any(cnt > QCAP - GRP)) drain();
            }
        }
        drain();
    }

    // scan one tile (all 64 points) for every lane whose bound allows it
    auto scan_tile = [&](int t, const float4& blo, const float4& bhi) {
        const float lb = box_bound(q.x, q.y, q.z, blo, bhi);
        const float limit = (t == my_tile) ? -1.0f : worst;   // own seed tile is already in the list
        if (!__any(lb <= limit)) return;
        pairs += PT;                                     // every lane evaluates the whole tile
        __builtin_amdgcn_wave_barrier();
        tile[lane] = sp[(size_t)t * PT + lane];          // one coalesced 1 KiB load per wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (int j0 = 0; j0 < PT; j0 += GRP) {
            float4 p[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) p[u] = tile[j0 + u];       // wave-uniform: LDS broadcast
            float d[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) d[u] = sqdist3(q.x, q.y, q.z, p[u].x, p[u].y, p[u].z);
            const float lim = (t == my_tile) ? -1.0f : worst;
#pragma unroll
            for (int u = 0; u < GRP; ++u) offer(d[u], __float_as_uint(p[u].w), lim);
            if constexpr (K > 1) {
                if (__any(cnt > QCAP - GRP)) drain();
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // 2) sweep: chunks of 64 tiles outwards from t0's chunk; lane l tests tile c*64+l against the
    //    wave's query box and the largest K-th distance in the wave
    const int nchunk = (nt + 63) >> 6;
    const int c0 = t0 >> 6;
    for (int s = 0; s < 2 * nchunk; ++s) {
        const int c = (s & 1) ? c0 - ((s + 1) >> 1) : c0 + (s >> 1);
        if (c < 0 || c >= nchunk) continue;
        const int t = (c << 6) + lane;
        float4 blo = make_float4(0, 0, 0, 0), bhi = blo;
        bool want = false;
        const float wmax = wave_max(worst);
        if (t < nt) {
            blo = bx[t * 2];
            bhi = bx[t * 2 + 1];
            want = boxbox_bound(qlo, qhi, blo, bhi) <= wmax;
        }
        unsigned long long mask = __ballot(want);
        while (mask) {
            const int l = (int)__ffsll((long long)mask) - 1;
            mask &= mask - 1;
            float4 lo4, hi4;
            lo4.x = lane_bcast(blo.x, l); lo4.y = lane_bcast(blo.y, l); lo4.z = lane_bcast(blo.z, l); lo4.w = 0.f;
            hi4.x = lane_bcast(bhi.x, l); hi4.y = lane_bcast(bhi.y, l); hi4.z = lane_bcast(bhi.z, l); hi4.w = 0.f;
            scan_tile((c << 6) + l, lo4, hi4);
        }
    }
    drain();
    publish_pairs(pairs);

    if (live) {
        const size_t o = ((size_t)b * Q + q_orig) * (size_t)Kout;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k < Kout) {
                const uint32_t id = (uint32_t)top.k[k];
                if (idx64) idx64[o + k] = (int64_t)id;
                if (idx32) idx32[o + k] = (int32_t)id;
                if (dist) dist[o + k] = __uint_as_float((uint32_t)(top.k[k] >> 32));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K <= 16: one ROW of 16 lanes (a DPP row, 4 rows per wave) owns one query.
//   * lane r of the row holds rank r of the running top-16 as a packed (distance, index) key;
//     an insert is a 16-lane ballot (how many entries sort before the candidate) plus one
//     row_shr:1 DPP shift -- no per-lane register lists, no LDS queues;
//   * the row tests 16 boxes at a time (one per lane): level-2 boxes first, then the 16 tile
//     boxes of every level-2 box that survives, then the 64 points of every surviving tile in
//     4 coalesced 256-byte row loads.  Every lane does useful work whether the queries of a
//     wave are neighbours in space or not, so sparse query sets (a few hundred points against
//     a 76800-pixel grid) cost the same per query as dense ones.
// grid = (ceil(Q_pad / 16), B)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t row_bits(unsigned long long ballot, int lane)
{
    return (uint32_t)(ballot >> (lane & 48)) & 0xffffu;
}
__device__ __forceinline__ u64 row_get(u64 v, int j)   // value of v in lane j of my row
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, j, 16);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), j, 16);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 row_shr1(u64 v)         // lane r gets lane r-1 (lane 0 keeps its own)
{
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x111, 0xf, 0xf, false);
    return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}

template <int K>
__device__ __forceinline__ void
knn_row16_body(const float4* __restrict__ spts, const float4* __restrict__ boxes,
               const float4* __restrict__ boxes2, const float* __restrict__ sframe,
               const uint32_t* __restrict__ skeys, int S, int S_pad, int nt, int nt2,
               const float4* __restrict__ qpts, const float* __restrict__ qraw, int Q, int Q_pad,
               int64_t* __restrict__ idx64, int32_t* __restrict__ idx32, float* __restrict__ dist,
               int Kout, const int blk_x, const int b)
{
    static_assert(K >= 2 && K <= 16, "row kernel holds one rank per lane");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = tid & 15;
    const int qslot = blk_x * (BLK / 16) + (tid >> 4);
    // queries either Morton-prepared (float4 with the original index in .w) or the caller's raw
    // [B,Q,3] array in its own order: a row works alone, so query order only affects cache locality
    float4 q;
    if (qraw) {
        const float* s = qraw + ((size_t)b * Q + min(qslot, Q - 1)) * 3;
        q = make_float4(s[0], s[1], s[2], __uint_as_float(qslot < Q ? (uint32_t)qslot : 0xffffffffu));
    } else {
        q = qpts[(size_t)b * Q_pad + min(qslot, Q_pad - 1)];
    }
    const uint32_t q_orig = __float_as_uint(q.w);
    const bool live = qslot < Q_pad && q_orig != 0xffffffffu;
    if (!__any(live)) return;

    const float4* sp = spts + (size_t)b * S_pad;
    const float4* bx = boxes + (size_t)b * nt * 2;
    const float4* bx2 = boxes2 + (size_t)b * nt2 * 2;

    u64 L = KEY_EMPTY;                       // rank r of the row's sorted top-16
    float worst = live ? FLT_MAX : -1.0f;    // row-uniform; -1 = this row never takes anything

    // offer the row's 16 candidate keys (one per lane, `pass` marks the useful ones)
    auto merge = [&](u64 key, bool pass) {
        uint32_t m = row_bits(__ballot(pass), lane);
        while (__any(m != 0)) {
            if (m != 0) {                                    // row-uniform
                const int j = __ffs((int)m) - 1;
                m &= m - 1;
                const u64 x = row_get(key, j);
                const int p = __popc(row_bits(__ballot(L < x), lane));   // entries sorting before x
                const u64 prev = row_shr1(L);
                L = (r < p) ? L : ((r == p) ? x : prev);
            }
        }
        worst = live ? __uint_as_float((uint32_t)(row_get(L, K - 1) >> 32)) : -1.0f;
    };

    unsigned int pairs = 0;
    auto scan_tile = [&](int t) {                            // t row-uniform
        const float4* tp = sp + (size_t)t * PT;
        pairs += PT / 16;
        float4 p[PT / 16];
#pragma unroll
        for (int c = 0; c < PT / 16; ++c) p[c] = tp[c * 16 + r];   // 4 x 256 B per row, all in flight
#pragma unroll
        for (int c = 0; c < PT / 16; ++c) {
            const float d = sqdist3(q.x, q.y, q.z, p[c].x, p[c].y, p[c].z);
            merge(pack_key(d, __float_as_uint(p[c].w)), d <= worst);
        }
    };

    // seed with the tile holding the query's own position on the support's Z-curve:
    // 17-ary lower_bound over the sorted keys, 16 probes per step (one per lane of the row)
    int my_tile = 0;
    {
        const uint32_t kq = morton_key(q.x, q.y, q.z, sframe + b * 8) & ~((1u << SORT_DROP) - 1u);      // the stored keys carry the sorted bits only
        const uint32_t* sk = skeys + (size_t)b * S_pad;
        int lo = 0, hi = live ? S : 0;
        while (__any(hi > lo)) {
            if (hi > lo) {                                   // row-uniform
                const long long span = hi - lo;
                const int pos = lo + (int)((span * (r + 1)) / 17);
                const int c = __popc(row_bits(__ballot(sk[pos] < kq), lane));   // probes below kq: 1..1 0..0
                const int new_lo = c > 0 ? lo + (int)((span * c) / 17) + 1 : lo;
                const int new_hi = c < 16 ? lo + (int)((span * (c + 1)) / 17) : hi;
                lo = new_lo;
                hi = new_hi;
            }
        }
        my_tile = min(lo / PT, nt - 1);
    }
    // the query's own tile and its two neighbours on the curve (a Z-curve neighbour on one
    // side can be far away in space, the one on the other side is then usually close)
    const int seed_lo = max(my_tile - 1, 0), seed_hi = min(my_tile + 1, nt - 1);
    if (live) {
        scan_tile(my_tile);
        if (seed_lo != my_tile) scan_tile(seed_lo);
        if (seed_hi != my_tile) scan_tile(seed_hi);
    }

    // level 2 -> level 1 -> points
    for (int g0 = 0; g0 < nt2; g0 += 16) {                   // uniform trip count
        const int g = g0 + r;
        bool want2 = false;
        if (g < nt2) want2 = box_bound(q.x, q.y, q.z, bx2[g * 2], bx2[g * 2 + 1]) <= worst;
        uint32_t m2 = row_bits(__ballot(want2), lane);
        while (__any(m2 != 0)) {
            if (m2 != 0) {
                const int gsel = g0 + __ffs((int)m2) - 1;     // row-uniform level-2 box
                m2 &= m2 - 1;
                const int t = gsel * FAN + r;
                float lb1 = INFINITY;                         // this lane's tile bound
                if (t < nt && (t < seed_lo || t > seed_hi)) lb1 = box_bound(q.x, q.y, q.z, bx[t * 2], bx[t * 2 + 1]);
                uint32_t m1 = row_bits(__ballot(lb1 <= worst), lane);
                while (m1 != 0) {                             // row-uniform loop
                    const int j = __ffs((int)m1) - 1;
                    m1 &= m1 - 1;
                    // the K-th distance may have tightened since the ballot: re-check the bound
                    // (held by lane j of the row) before paying for the tile
                    const float lbj = __shfl(lb1, j, 16);
                    if (lbj <= worst) scan_tile(gsel * FAN + j);
                }
            }
        }
    }

    publish_pairs(pairs);
    if (live && r < Kout) {
        const size_t o = ((size_t)b * Q + q_orig) * (size_t)Kout + r;
        const uint32_t id = (uint32_t)L;
        if (idx64) idx64[o] = (int64_t)id;
        if (idx32) idx32[o] = (int32_t)id;
        if (dist) dist[o] = __uint_as_float((uint32_t)(L >> 32));
    }
}

// ---- kernels: one search per launch (grid = (query blocks, B)), or several searches in ONE launch -- the searches of an
// index pyramid read only the cloud and the xyz image, none depends on another (linemod_dataset.py:299-353), so all searches
// that run the same kernel go out together: blockIdx.x walks the (search, frame, query block) triples of a table in the
// kernel arguments.
struct PrunedArgs {
    const float4 *spts, *boxes, *boxes2, *qpts;
    const uint32_t *skeys, *scell;
    const float *sframe, *qraw;
    int64_t* idx64;
    int32_t* idx32;
    float* dist;
    int S, S_pad, nt, nt2, Q, Q_pad, Kout;
    int gx;               // query blocks per frame
    int blk0;             // first block of this search in the flattened grid
};
constexpr int MAX_SEARCHES = 12;
struct MultiPruned {
    PrunedArgs a[MAX_SEARCHES];
    int n;
};

// search of flattened block `blk`: the last one whose first block is <= blk
__device__ __forceinline__ int find_search(const MultiPruned& m, int blk)
{
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_SEARCHES; ++i) s += (i < m.n && m.a[i].blk0 <= blk) ? 1 : 0;
    return s;
}

template <int K>
__global__ void __launch_bounds__(BLK)
knn_pruned_kernel(const float4* __restrict__ spts, const float4* __restrict__ boxes,
                  const uint32_t* __restrict__ skeys, const float* __restrict__ sframe,
                  const uint32_t* __restrict__ scell, int S, int S_pad, int nt,
                  const float4* __restrict__ qpts, int Q, int Q_pad,
                  int64_t* __restrict__ idx64, int32_t* __restrict__ idx32, float* __restrict__ dist,
                  int Kout)
{
    knn_pruned_body<K>(spts, boxes, skeys, sframe, scell, S, S_pad, nt, qpts, Q, Q_pad, idx64, idx32, dist, Kout, blockIdx.x, blockIdx.y);
}

template <int K>
__global__ void __launch_bounds__(BLK)
knn_pruned_multi_kernel(const MultiPruned m)
{
    const PrunedArgs& a = m.a[find_search(m, blockIdx.x)];
    const int local = blockIdx.x - a.blk0;
    knn_pruned_body<K>(a.spts, a.boxes, a.skeys, a.sframe, a.scell, a.S, a.S_pad, a.nt, a.qpts, a.Q, a.Q_pad, a.idx64, a.idx32, a.dist,
                       a.Kout, local % a.gx, local / a.gx);
}

template <int K>
__global__ void __launch_bounds__(BLK)
knn_row16_kernel(const float4* __restrict__ spts, const float4* __restrict__ boxes,
                 const float4* __restrict__ boxes2, const float* __restrict__ sframe,
                 const uint32_t* __restrict__ skeys, int S, int S_pad, int nt, int nt2,
                 const float4* __restrict__ qpts, const float* __restrict__ qraw, int Q, int Q_pad,
                 int64_t* __restrict__ idx64, int32_t* __restrict__ idx32, float* __restrict__ dist,
                 int Kout)
{
    knn_row16_body<K>(spts, boxes, boxes2, sframe, skeys, S, S_pad, nt, nt2, qpts, qraw, Q, Q_pad, idx64, idx32, dist, Kout,
                      blockIdx.x, blockIdx.y);
}

template <int K>
__global__ void __launch_bounds__(BLK)
knn_row16_multi_kernel(const MultiPruned m)
{
    const PrunedArgs& a = m.a[find_search(m, blockIdx.x)];
    const int local = blockIdx.x - a.blk0;
    knn_row16_body<K>(a.spts, a.boxes, a.boxes2, a.sframe, a.skeys, a.S, a.S_pad, a.nt, a.nt2, a.qpts, a.qraw, a.Q, a.Q_pad, a.idx64,
                      a.idx32, a.dist, a.Kout, local % a.gx, local / a.gx);
}

int pad_k(int K)
{
    int p = 1;
    while (p < K) p <<= 1;
    return p;
}

size_t sort_temp_bound(size_t n, size_t sort_blocks)
{
    // the hand-written segmented sort (csrc/seg_sort.hip): one 256-bin histogram per workgroup; rocPRIM (64-bit keys only): two
    // alternate key/value buffers + histograms/look-back state
    return std::max(n * (sizeof(unsigned long long) + sizeof(uint32_t)) + (size_t)(4u << 20), sort_blocks * 256 * sizeof(uint32_t));
}

struct PrepWs {
    size_t keys_in, keys_out, vals_in, vals_out, bbox, temp, temp_bytes, total;
};

PrepWs prep_ws_n(size_t n, size_t sort_blocks);
PrepWs prep_ws(int64_t B, int64_t S) { return prep_ws_n((size_t)B * S, (size_t)B * (size_t)ceil_div(S, segsort::CHUNK)); }

// One stable sort of all (segment, truncated Morton key) pairs; `plan` = the (set, frame) segments as ranges of the key array.
// 32-bit keys: the segmented radix sort of csrc/seg_sort.hip on the SORT_BITS key bits (segments never mix: the segment number in the
// key's high bits is carried along, not sorted on) -- 6 launches; round 4 called rocprim::radix_sort_pairs here, which ran its merge
// sort for these sizes: 20 launches, 0.13 ms alone, 0.40 ms inside the bench step.  64-bit keys (more than 256 segments): rocPRIM.
constexpr int64_t MAX_SEG32 = 1LL << (32 - SORT_BITS);
int sort_segments(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t, int64_t, segsort::Plan& plan,
                  void* temp, size_t temp_bytes, hipStream_t st, const char* who)
{
    bool in_alt = false;
    const hipError_t e = segsort::sort_pairs(plan, keys_in, vals_in, keys_out, vals_out, SORT_BITS, temp, temp_bytes, st, &in_alt);
    if (e != hipSuccess) return set_error(e == hipErrorInvalidValue ? FFB6D_ERR_WORKSPACE : FFB6D_ERR_HIP, "%s: segmented sort failed: %s", who,
                                          hipGetErrorString(e));
    static_assert((SORT_BITS + 7) / 8 % 2 == 1, "an odd number of passes leaves the result in keys_out / vals_out");
    return in_alt ? FFB6D_OK : set_error(FFB6D_ERR_HIP, "%s: segmented sort left its result in the wrong buffer", who);
}

int sort_segments(unsigned long long* keys_in, unsigned long long* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, int64_t segments,
                  segsort::Plan&, void* temp, size_t temp_bytes, hipStream_t st, const char* who)
{
    unsigned end_bit = SORT_BITS;
    while ((1LL << (end_bit - SORT_BITS)) < segments) ++end_bit;
    size_t need = 0;
    FFB6D_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, st));
    if (need > temp_bytes) return set_error(FFB6D_ERR_WORKSPACE, "%s: radix sort wants %zu temp bytes, reserved %zu", who, need, temp_bytes);
    size_t have = temp_bytes;
    FFB6D_HIP_TRY(rocprim::radix_sort_pairs(temp, have, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, st));
    return FFB6D_OK;
}

PrepWs prep_ws_n(size_t n, size_t sort_blocks)
{
    PrepWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    w.keys_in = take(n * 8);
    w.keys_out = take(n * 8);
    w.vals_in = take(n * 4);
    w.vals_out = take(n * 4);
    w.bbox = take(4096);                       /* unused (frames live in the prepared blob) */
    w.temp_bytes = sort_temp_bound(n, sort_blocks);
    w.temp = take(w.temp_bytes);
    w.total = o;
    return w;
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

extern "C" {

size_t ffb6d_knn_prepared_bytes(int64_t B, int64_t S)
{
    if (B <= 0 || S <= 0) return 0;
    return layout(B, S).total;
}

size_t ffb6d_knn_prepare_workspace_bytes(int64_t B, int64_t S)
{
    if (B <= 0 || S <= 0) return 0;
    return prep_ws(B, S).total;
}

int ffb6d_knn_prepare(const float* pts, int64_t B, int64_t S, void* prepared, size_t prepared_bytes,
                      void* workspace, size_t workspace_bytes, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 1 && S >= 1, "knn_prepare: empty point set");
    FFB6D_REQUIRE(S < (1LL << 31) && B < 65536, "knn_prepare: size too large");
    FFB6D_REQUIRE(pts && prepared && workspace, "knn_prepare: null pointer");
    const Layout L = layout(B, S);
    const PrepWs W = prep_ws(B, S);
    if (prepared_bytes < L.total || workspace_bytes < W.total)
        return set_error(FFB6D_ERR_WORKSPACE, "knn_prepare: need %zu prepared + %zu workspace bytes, got %zu + %zu",
                         L.total, W.total, prepared_bytes, workspace_bytes);
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    char* pp = static_cast<char*>(prepared);
    auto* keys_in = reinterpret_cast<unsigned long long*>(ws + W.keys_in);
    auto* keys_out = reinterpret_cast<unsigned long long*>(ws + W.keys_out);
    auto* vals_in = reinterpret_cast<uint32_t*>(ws + W.vals_in);
    auto* vals_out = reinterpret_cast<uint32_t*>(ws + W.vals_out);
    float* frame = reinterpret_cast<float*>(pp + L.frame_off);

    hipLaunchKernelGGL(frame_kernel, dim3((unsigned)B), dim3(1024), 0, st, pts, (int)S, frame);
    const size_t n = (size_t)B * S;
    auto run = [&](auto* kin, auto* kout) -> int {
        using KeyT = std::remove_pointer_t<decltype(kin)>;
        hipLaunchKernelGGL((morton_kernel<KeyT>), dim3((unsigned)ceil_div(S, BLK), (unsigned)B), dim3(BLK), 0, st, pts, (int)S, frame, kin,
                           vals_in);
        FFB6D_LAUNCH_CHECK();
        segsort::Plan plan;
        plan.ngroups = 1; plan.B = (int)B; plan.g[0].pos0 = 0; plan.g[0].S = (int)S;
        if (const int rc = sort_segments(kin, kout, vals_in, vals_out, n, B, plan, ws + W.temp, W.temp_bytes, st, "knn_prepare")) return rc;
        hipLaunchKernelGGL((gather_box_kernel<KeyT>), dim3((unsigned)ceil_div(L.nt, BLK / 64), (unsigned)B), dim3(BLK), 0, st, pts,
                           (int)S, (int)L.S_pad, (int)L.nt, kout, vals_out, reinterpret_cast<float4*>(pp + L.pts_off),
                           reinterpret_cast<float4*>(pp + L.box_off), reinterpret_cast<uint32_t*>(pp + L.key_off));
        return FFB6D_OK;
    };
    const int rc = B <= MAX_SEG32 ? run(reinterpret_cast<uint32_t*>(keys_in), reinterpret_cast<uint32_t*>(keys_out)) : run(keys_in, keys_out);
    if (rc) return rc;
    hipLaunchKernelGGL(box2_kernel, dim3((unsigned)ceil_div(L.nt2, BLK), (unsigned)B), dim3(BLK), 0, st,
                       reinterpret_cast<const float4*>(pp + L.box_off), (int)L.nt, (int)L.nt2,
                       reinterpret_cast<float4*>(pp + L.box2_off));
    hipLaunchKernelGGL(cell_table_kernel, dim3(NCELL / BLK, (unsigned)B), dim3(BLK), 0, st,
                       reinterpret_cast<const uint32_t*>(pp + L.key_off), (int)S, (int)L.S_pad, (int)L.nt,
                       reinterpret_cast<uint32_t*>(pp + L.cell_off));
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

size_t ffb6d_knn_prepare_multi_workspace_bytes(int nsets, const int64_t* npts, int64_t B)
{
    if (nsets <= 0 || !npts || B <= 0) return 0;
    size_t n = 0, blocks = 0;
    for (int i = 0; i < nsets; ++i) {
        const size_t s = (size_t)(npts[i] > 0 ? npts[i] : 0);
        n += (size_t)B * s;
        blocks += (size_t)B * (size_t)ceil_div((int64_t)s, segsort::CHUNK);
    }
    return prep_ws_n(n, blocks).total;
}

int ffb6d_knn_prepare_multi(int nsets, const float* const* pts, const int64_t* npts, int64_t B, void* const* prepared,
                            const size_t* prepared_bytes, void* workspace, size_t workspace_bytes, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(nsets >= 1 && nsets <= MAX_SETS, "knn_prepare_multi: 1..%d sets per call (got %d)", MAX_SETS, nsets);
    FFB6D_REQUIRE(pts && npts && prepared && prepared_bytes && workspace, "knn_prepare_multi: null pointer");
    FFB6D_REQUIRE(B >= 1 && B * (int64_t)nsets < 65536, "knn_prepare_multi: bad batch size");
    MultiDesc m;
    m.B = (int)B;
    size_t n = 0;
    int64_t S_max = 0, nt_max = 0, nt2_max = 0;
    for (int i = 0; i < nsets; ++i) {
        FFB6D_REQUIRE(npts[i] >= 1 && npts[i] < (1LL << 31) && pts[i] && prepared[i], "knn_prepare_multi: set %d is empty or null", i);
        const Layout L = layout(B, npts[i]);
        if (prepared_bytes[i] < L.total)
            return set_error(FFB6D_ERR_WORKSPACE, "knn_prepare_multi: set %d needs %zu prepared bytes, got %zu", i, L.total, prepared_bytes[i]);
        char* pp = static_cast<char*>(prepared[i]);
        SetDesc& d = m.s[i];
        d.pts = pts[i];
        d.out_pts = reinterpret_cast<float4*>(pp + L.pts_off);
        d.boxes = reinterpret_cast<float4*>(pp + L.box_off);
        d.boxes2 = reinterpret_cast<float4*>(pp + L.box2_off);
        d.out_keys = reinterpret_cast<uint32_t*>(pp + L.key_off);
        d.cell = reinterpret_cast<uint32_t*>(pp + L.cell_off);
        d.frame = reinterpret_cast<float*>(pp + L.frame_off);
        d.S = (int)npts[i]; d.S_pad = (int)L.S_pad; d.nt = (int)L.nt; d.nt2 = (int)L.nt2;
        d.pos0 = (long long)n;
        n += (size_t)B * (size_t)npts[i];
        S_max = std::max<int64_t>(S_max, npts[i]); nt_max = std::max<int64_t>(nt_max, L.nt); nt2_max = std::max<int64_t>(nt2_max, L.nt2);
    }
    segsort::Plan plan;
    plan.ngroups = nsets; plan.B = (int)B;
    size_t sort_blocks = 0;
    for (int i = 0; i < nsets; ++i) {
        plan.g[i].pos0 = m.s[i].pos0; plan.g[i].S = m.s[i].S;
        sort_blocks += (size_t)B * (size_t)ceil_div(npts[i], segsort::CHUNK);
    }
    const PrepWs W = prep_ws_n(n, sort_blocks);
    if (workspace_bytes < W.total)
        return set_error(FFB6D_ERR_WORKSPACE, "knn_prepare_multi: need %zu workspace bytes, got %zu", W.total, workspace_bytes);
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    auto* keys_in = reinterpret_cast<unsigned long long*>(ws + W.keys_in);
    auto* keys_out = reinterpret_cast<unsigned long long*>(ws + W.keys_out);
    auto* vals_in = reinterpret_cast<uint32_t*>(ws + W.vals_in);
    auto* vals_out = reinterpret_cast<uint32_t*>(ws + W.vals_out);
    const unsigned ns = (unsigned)nsets, nb = (unsigned)B;
    hipLaunchKernelGGL(frame_multi_kernel, dim3(nb, ns), dim3(1024), 0, st, m);
    auto run = [&](auto* kin, auto* kout) -> int {
        using KeyT = std::remove_pointer_t<decltype(kin)>;
        hipLaunchKernelGGL((morton_multi_kernel<KeyT>), dim3((unsigned)ceil_div(S_max, BLK), nb, ns), dim3(BLK), 0, st, m, kin, vals_in);
        FFB6D_LAUNCH_CHECK();
        if (const int rc = sort_segments(kin, kout, vals_in, vals_out, n, B * nsets, plan, ws + W.temp, W.temp_bytes, st, "knn_prepare_multi"))
            return rc;
        hipLaunchKernelGGL((gather_box_multi_kernel<KeyT>), dim3((unsigned)ceil_div(nt_max, BLK / 64), nb, ns), dim3(BLK), 0, st, m, kout,
                           vals_out);
        return FFB6D_OK;
    };
    const int rc = B * nsets <= MAX_SEG32 ? run(reinterpret_cast<uint32_t*>(keys_in), reinterpret_cast<uint32_t*>(keys_out))
                                          : run(keys_in, keys_out);
    if (rc) return rc;
    hipLaunchKernelGGL(box2_multi_kernel, dim3((unsigned)ceil_div(nt2_max, BLK), nb, ns), dim3(BLK), 0, st, m);
    hipLaunchKernelGGL(cell_table_multi_kernel, dim3(NCELL / BLK, nb, ns), dim3(BLK), 0, st, m);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_knn_search_prepared(const void* prep_support, const void* prep_query, const float* raw_query,
                              int64_t B, int64_t S, int64_t Q, int K, int64_t* idx64, int32_t* idx32,
                              float* dist, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(K >= 1 && K <= 32, "knn_search_prepared: K must be in [1,32] (got %d)", K);
    FFB6D_REQUIRE(B >= 1 && S >= 1 && Q >= 1, "knn_search_prepared: empty problem");
    FFB6D_REQUIRE(S >= K, "knn_search_prepared: npts (%lld) < K (%d)", (long long)S, K);
    FFB6D_REQUIRE(prep_support && (prep_query || raw_query), "knn_search_prepared: null point set");
    FFB6D_REQUIRE(prep_query || (K >= 2 && K <= 16),
                  "knn_search_prepared: raw (unprepared) queries are supported for 2 <= K <= 16 only");
    FFB6D_REQUIRE(idx64 || idx32 || dist, "knn_search_prepared: no output requested");
    const Layout LS = layout(B, S), LQ = layout(B, Q);
    const char* ps = static_cast<const char*>(prep_support);
    const char* pq = static_cast<const char*>(prep_query);
    const float4* spts = reinterpret_cast<const float4*>(ps + LS.pts_off);
    const float4* boxes = reinterpret_cast<const float4*>(ps + LS.box_off);
    const uint32_t* skeys = reinterpret_cast<const uint32_t*>(ps + LS.key_off);
    const float* sframe = reinterpret_cast<const float*>(ps + LS.frame_off);
    const uint32_t* scell = reinterpret_cast<const uint32_t*>(ps + LS.cell_off);
    const float4* qpts = prep_query ? reinterpret_cast<const float4*>(pq + LQ.pts_off) : nullptr;
    dim3 grid((unsigned)ceil_div(LQ.S_pad, BLK), (unsigned)B);
    hipStream_t st = as_stream(stream);
    const int Kp = pad_k(K);
    FFB6D_REQUIRE(K <= 16 || prep_query, "knn_search_prepared: K > 16 needs prepared queries");
    if (K >= 2 && K <= 16) {   // row-cooperative kernel: 16 lanes per query
        const float4* boxes2 = reinterpret_cast<const float4*>(ps + LS.box2_off);
        dim3 rgrid((unsigned)ceil_div(LQ.S_pad, BLK / 16), (unsigned)B);
#define FFB6D_LAUNCH_ROW(KK)                                                                                  \
    hipLaunchKernelGGL((knn_row16_kernel<KK>), rgrid, dim3(BLK), 0, st, spts, boxes, boxes2, sframe, skeys,    \
                       (int)S, (int)LS.S_pad, (int)LS.nt, (int)LS.nt2, qpts, prep_query ? nullptr : raw_query, (int)Q,  \
                       (int)LQ.S_pad, idx64, idx32, dist, K)
        switch (Kp) {
            case 2: FFB6D_LAUNCH_ROW(2); break;
            case 4: FFB6D_LAUNCH_ROW(4); break;
            case 8: FFB6D_LAUNCH_ROW(8); break;
            default: FFB6D_LAUNCH_ROW(16); break;
        }
#undef FFB6D_LAUNCH_ROW
        FFB6D_LAUNCH_CHECK();
        return FFB6D_OK;
    }
    const size_t lds = (size_t)(BLK / 64) * PT * sizeof(float4) + (Kp > 1 ? (size_t)QCAP * BLK * sizeof(uint2) : 0);
#define FFB6D_LAUNCH_PRUNED(KK)                                                                              \
    hipLaunchKernelGGL((knn_pruned_kernel<KK>), grid, dim3(BLK), lds, st, spts, boxes, skeys, sframe, scell, (int)S, \
                       (int)LS.S_pad, (int)LS.nt, qpts, (int)Q, (int)LQ.S_pad, idx64, idx32, dist, K)
    switch (Kp) {
        case 1: FFB6D_LAUNCH_PRUNED(1); break;
        case 2: FFB6D_LAUNCH_PRUNED(2); break;
        case 4: FFB6D_LAUNCH_PRUNED(4); break;
        case 8: FFB6D_LAUNCH_PRUNED(8); break;
        case 16: FFB6D_LAUNCH_PRUNED(16); break;
        case 32: FFB6D_LAUNCH_PRUNED(32); break;
        default: return set_error(FFB6D_ERR_ARG, "knn_search_prepared: unsupported K=%d", K);
    }
#undef FFB6D_LAUNCH_PRUNED
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // extern "C"

namespace ffb6d {
// Searches of `s` (indices `which`, all routed to the Morton-ordered kernels) in as few launches as kernels: one per padded K
// of the 16-lane row kernel, one for K = 1.  Called by ffb6d_knn_search_multi (csrc/knn.hip).
int knn_search_multi_prepared(const ffb6d_knn_search_t* s, const int* which, int n, int64_t B, hipStream_t st)
{
    const int kps[5] = {1, 2, 4, 8, 16};
    for (int kp : kps) {
        MultiPruned m;
        m.n = 0;
        int blocks = 0;
        auto flush = [&]() -> int {
            if (m.n == 0) return FFB6D_OK;
            const size_t lds = (size_t)(BLK / 64) * PT * sizeof(float4);
            switch (kp) {
                case 1: hipLaunchKernelGGL((knn_pruned_multi_kernel<1>), dim3((unsigned)blocks), dim3(BLK), lds, st, m); break;
                case 2: hipLaunchKernelGGL((knn_row16_multi_kernel<2>), dim3((unsigned)blocks), dim3(BLK), 0, st, m); break;
                case 4: hipLaunchKernelGGL((knn_row16_multi_kernel<4>), dim3((unsigned)blocks), dim3(BLK), 0, st, m); break;
                case 8: hipLaunchKernelGGL((knn_row16_multi_kernel<8>), dim3((unsigned)blocks), dim3(BLK), 0, st, m); break;
                default: hipLaunchKernelGGL((knn_row16_multi_kernel<16>), dim3((unsigned)blocks), dim3(BLK), 0, st, m); break;
            }
            FFB6D_LAUNCH_CHECK();
            m.n = 0;
            blocks = 0;
            return FFB6D_OK;
        };
        for (int w = 0; w < n; ++w) {
            const ffb6d_knn_search_t& q = s[which[w]];
            if (pad_k(q.K) != kp) continue;
            FFB6D_REQUIRE(q.prep_support && (q.prep_query || (q.query && q.K >= 2)), "knn_search_multi: search %d needs a prepared support "
                          "and prepared (K = 1) or raw (K >= 2) queries", which[w]);
            const Layout LS = layout(B, q.S), LQ = layout(B, q.Q);
            const char* ps = static_cast<const char*>(q.prep_support);
            const char* pq = static_cast<const char*>(q.prep_query);
            PrunedArgs& a = m.a[m.n];
            a.spts = reinterpret_cast<const float4*>(ps + LS.pts_off);
            a.boxes = reinterpret_cast<const float4*>(ps + LS.box_off);
            a.boxes2 = reinterpret_cast<const float4*>(ps + LS.box2_off);
            a.skeys = reinterpret_cast<const uint32_t*>(ps + LS.key_off);
            a.sframe = reinterpret_cast<const float*>(ps + LS.frame_off);
            a.scell = reinterpret_cast<const uint32_t*>(ps + LS.cell_off);
            a.qpts = pq ? reinterpret_cast<const float4*>(pq + LQ.pts_off) : nullptr;
            a.qraw = pq ? nullptr : q.query;
            a.idx64 = q.idx64; a.idx32 = q.idx32; a.dist = q.dist;
            a.S = (int)q.S; a.S_pad = (int)LS.S_pad; a.nt = (int)LS.nt; a.nt2 = (int)LS.nt2;
            a.Q = (int)q.Q; a.Q_pad = (int)LQ.S_pad; a.Kout = q.K;
            a.gx = (int)ceil_div(LQ.S_pad, kp == 1 ? BLK : BLK / 16);
            a.blk0 = blocks;
            blocks += a.gx * (int)B;
            if (++m.n == MAX_SEARCHES) {
                const int rc = flush();
                if (rc != FFB6D_OK) return rc;
            }
        }
        const int rc = flush();
        if (rc != FFB6D_OK) return rc;
    }
    return FFB6D_OK;
}
}  // namespace ffb6d

extern "C" int ffb6d_knn_set_pair_counter(unsigned long long* device_counter)
{
    FFB6D_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ffb6d::g_pair_counter), &device_counter, sizeof(device_counter)));
    return FFB6D_OK;
}
