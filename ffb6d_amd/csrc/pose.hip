// ffb6d_amd/csrc/pose.hip -- gfx950 pose solver (include/ffb6d_pose.h).
//
// The reference runs MeanShiftTorch.fit (ffb6d/utils/meanshift_pytorch.py:27-58) once per object
// centre and once per keypoint, each call materialising [M,M,3] / [M,M] temporaries per round and
// synchronising with the host for the stopping test.  Here every (frame, object, keypoint) vote set
// of a batch is one row of a [G, set_stride] float4 table and one launch advances all of them by one
// round:
//   * a 256-thread block owns 64 points of one set; 4 lanes share a point and each takes a quarter
//     of the set, which is streamed through LDS in 512-point tiles (same-address LDS reads
//     broadcast, 4 distinct addresses per wavefront -> conflict free);
//   * the Gaussian weight is one v_exp_f32: exp(-0.5 (d/bw)^2) = 2^(d^2 * k), k = -0.5 log2(e)/bw^2;
//     the normalising constant of the reference's kernel cancels in the weighted mean;
//   * the stopping test stays on the device: round t publishes its largest move with atomicMax
//     into slot t%3, reads slot (t-1)%3 to know whether its set already stopped (then the launch is
//     a no-op for that set) and clears slot (t+1)%3.
// fp32 throughout like the reference; results differ from it by summation order and exp rounding
// only (tests/test_pose_gpu.py states the tolerance).
#include <algorithm>
#include <vector>

#include "common.h"
#include "ffb6d_pose.h"

namespace {

using ffb6d::ceil_div;

constexpr int kBlock = 256;
constexpr int kLanesPerPoint = 4;
constexpr int kPointsPerBlock = kBlock / kLanesPerPoint;   // 64
constexpr int kTile = 512;
constexpr float kFar = 1.0e15f;                            // padding point: weight underflows to 0

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// workspace tail: shift [3][G] float bits of the largest move of round t in slot t%3; rounds [G]
// rounds made; best [G] (ball size << 32) | (0xffffffff - point index)

__device__ __forceinline__ void load_tile(float4* tile, const float4* src, int j0, int cnt) {
    for (int x = threadIdx.x; x < kTile; x += kBlock) {
        const int j = j0 + x;
        tile[x] = j < cnt ? src[j] : make_float4(kFar, kFar, kFar, 0.f);
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// one mean-shift round for every set (meanshift_pytorch.py:38-46).  The pair loop is written on
// 2-wide vectors so that it compiles to packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two pairs per
// instruction); the tile is kept as three planes in LDS so a lane fetches two neighbours per b64 read.
__global__ __launch_bounds__(kBlock) void mean_shift_round_kernel(
    float4* __restrict__ buf0, float4* __restrict__ buf1, const int* __restrict__ counts, int sets_per_count,
    int64_t stride, float k2, float thresh, unsigned* __restrict__ shift, int* __restrict__ rounds, int t, int G, int min_cnt,
    const int* __restrict__ handed) {
#pragma clang fp contract(fast)
    __shared__ __attribute__((aligned(16))) float tx[kTile], ty[kTile], tz[kTile];
    __shared__ float wmax[kBlock / 64];
    const int g = blockIdx.y;
    const int cnt = counts[g / sets_per_count];
    if (cnt <= min_cnt || (handed && handed[g] <= min_cnt)) return;     // fitted by mean_shift_fit_kernel (directly / after the chip-wide rounds)
    bool done = false;
    if (t > 0) done = __uint_as_float(shift[((t + 2) % 3) * G + g]) < thresh;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        shift[((t + 1) % 3) * G + g] = 0u;
        if (!done && cnt > 0) rounds[g] = t + 1;
    }
    const int q0 = blockIdx.x * kPointsPerBlock;
    if (done || q0 >= cnt) return;
    const float4* src = ((t & 1) ? buf1 : buf0) + g * stride;
    float4* dst = ((t & 1) ? buf0 : buf1) + g * stride;
    const int sub = threadIdx.x & (kLanesPerPoint - 1);
    const int q = q0 + (threadIdx.x >> 2);
    const float4 c = src[min(q, cnt - 1)];
    const v2f cx = {c.x, c.x}, cy = {c.y, c.y}, cz = {c.z, c.z}, kk = {k2, k2};
    v2f sw = {0.f, 0.f}, sx = {0.f, 0.f}, sy = {0.f, 0.f}, sz = {0.f, 0.f};
    for (int j0 = 0; j0 < cnt; j0 += kTile) {
        __syncthreads();
        for (int x = threadIdx.x; x < kTile; x += kBlock) {
            const int j = j0 + x;
            const float4 p = j < cnt ? src[j] : make_float4(kFar, kFar, kFar, 0.f);
            tx[x] = p.x;
            ty[x] = p.y;
            tz[x] = p.z;
        }
        __syncthreads();
        const int n = min(kTile, (cnt - j0 + 7) & ~7);
#pragma unroll 2
        for (int x = 2 * sub; x < n; x += 2 * kLanesPerPoint) {
            const v2f px = *reinterpret_cast<const v2f*>(&tx[x]);
            const v2f py = *reinterpret_cast<const v2f*>(&ty[x]);
            const v2f pz = *reinterpret_cast<const v2f*>(&tz[x]);
            const v2f dx = px - cx, dy = py - cy, dz = pz - cz;
            const v2f e = (dx * dx + dy * dy + dz * dz) * kk;
            v2f w;
            w.x = __builtin_amdgcn_exp2f(e.x);
            w.y = __builtin_amdgcn_exp2f(e.y);
            sw += w;
            sx += w * px;
            sy += w * py;
            sz += w * pz;
        }
    }
    float aw = sw.x + sw.y, ax = sx.x + sx.y, ay = sy.x + sy.y, az = sz.x + sz.y;
#pragma unroll
    for (int o = 1; o < kLanesPerPoint; o <<= 1) {
        aw += __shfl_xor(aw, o);
        ax += __shfl_xor(ax, o);
        ay += __shfl_xor(ay, o);
        az += __shfl_xor(az, o);
    }
    float move = 0.f;
    if (q < cnt) {
        const float nx = ax / aw, ny = ay / aw, nz = az / aw;
        if (sub == 0) dst[q] = make_float4(nx, ny, nz, c.w);
        const float ex = nx - c.x, ey = ny - c.y, ez = nz - c.z;
        move = sqrtf(ex * ex + ey * ey + ez * ez);
        if (!(move == move)) move = __uint_as_float(0x7f800000u);   // NaN never counts as converged
    }
    move = wave_max(move);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = move;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wmax[0];
        for (int i = 1; i < kBlock / 64; ++i) m = fmaxf(m, wmax[i]);
        atomicMax(&shift[(t % 3) * G + g], __float_as_uint(m));
    }
}

// One point of the one-workgroup fit against the positions [0, n): 4 lanes per point, lane `sub` takes the pairs x = 2 sub, 2 sub + 8, ...
// (two per step), then the four lanes meet.  aw / ax / ay / az: sums of w and w * position, am: multiplicity at the point's own
// position, r: lowest index there.  Shared by the fit kernel and by the kernel that spreads a set's first rounds over the chip: the same
// instructions in the same order, hence the same bits.
__device__ __forceinline__ void fit_pair_sums(const float* cx_, const float* cy_, const float* cz_, const float* pm, int n, int sub, v2f kk,
                                              v2f cx, v2f cy, v2f cz, float& aw, float& ax_, float& ay_, float& az_, float& am_, int& r)
{
#pragma clang fp contract(fast)
    v2f sw = {0.f, 0.f}, sx = {0.f, 0.f}, sy = {0.f, 0.f}, sz = {0.f, 0.f}, mult = {0.f, 0.f};
    int r0 = 1 << 30, r1 = 1 << 30;                   // (a point always finds itself: the sentinel never survives)
#pragma unroll 2
    for (int x = 2 * sub; x < n; x += 8) {
        const v2f ax = *reinterpret_cast<const v2f*>(&cx_[x]);
        const v2f ay = *reinterpret_cast<const v2f*>(&cy_[x]);
        const v2f az = *reinterpret_cast<const v2f*>(&cz_[x]);
        const v2f am = *reinterpret_cast<const v2f*>(&pm[x]);
        const v2f dx = ax - cx, dy = ay - cy, dz = az - cz;
        const v2f d2 = dx * dx + dy * dy + dz * dz;
        const v2f e = d2 * kk;
        v2f w;
        w.x = __builtin_amdgcn_exp2f(e.x);
        w.y = __builtin_amdgcn_exp2f(e.y);
        w *= am;
        sw += w;
        sx += w * ax;
        sy += w * ay;
        sz += w * az;
        const bool h0 = d2.x == 0.f, h1 = d2.y == 0.f;   // the same position (a far padding point never is)
        mult.x += h0 ? am.x : 0.f;
        mult.y += h1 ? am.y : 0.f;
        r0 = h0 ? min(r0, x) : r0;
        r1 = h1 ? min(r1, x + 1) : r1;
    }
    aw = sw.x + sw.y; ax_ = sx.x + sx.y; ay_ = sy.x + sy.y; az_ = sz.x + sz.y; am_ = mult.x + mult.y;
    r = min(r0, r1);
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
        aw += __shfl_xor(aw, o);
        ax_ += __shfl_xor(ax_, o);
        ay_ += __shfl_xor(ay_, o);
        az_ += __shfl_xor(az_, o);
        am_ += __shfl_xor(am_, o);
        r = min(r, __shfl_xor(r, o));
    }
}

// ---- the first two rounds of a set spread over the chip (round 6) -------------------------------------------------
// A fit is a chain of ~260 rounds in ONE workgroup, and its first rounds are the whole cost of the early collapse: round 0 of a
// 1700-vote set is 3 M weighted pairs = 1.25 ms on one CU, round 1 0.45 ms, the other ~260 rounds 3.3 us each
// (profiles/r06_pose_rounds_probe.txt) -- while a call with 40 sets (the centre votes of a batch) leaves 216 CUs idle.  This kernel
// makes ONE round for 128 points of a set per workgroup (the set in LDS as the fit keeps it, the fit's own pair function): new
// positions, multiplicity and representative of every point (one float4) and the set's largest move (atomic max), to global memory --
// into the round-by-round path's two position buffers and move slots, idle while the fits run.  ffb6d_mean_shift_f32 runs it twice
// (round 0 on the votes, round 1 on round 0's positions); the fit kernel then STARTS from those results -- round 1's only when round 0
// found no duplicate and did not converge, which is when its own round 1 would have read exactly these positions with multiplicity 1.
// Same arithmetic per point as the fit's own rounds: equal bits (tests/test_pose_gpu.py compares the two paths).
constexpr int kSpreadMin = 512;               // smaller sets: the fit's own first rounds are cheap
constexpr int kSpreadBT = 512;

__global__ __launch_bounds__(kSpreadBT) void mean_shift_spread_round_kernel(
    const float4* __restrict__ in, const int* __restrict__ counts, int sets_per_count, int64_t stride, int cap, float k2,
    float4* __restrict__ out, unsigned* __restrict__ move_of) {
#pragma clang fp contract(fast)
    extern __shared__ __attribute__((aligned(16))) unsigned char spread_lds[];
    float* px = reinterpret_cast<float*>(spread_lds);
    float *py = px + cap, *pz = py + cap, *pm = pz + cap;
    const int g = blockIdx.y;
    const int M = counts[g / sets_per_count];
    const int q0 = blockIdx.x * (kSpreadBT / 4);
    if (M < kSpreadMin || M > cap || q0 >= M) return;
    const int tid = threadIdx.x, sub = tid & 3;
    const int n = (M + 7) & ~7;
    for (int j = tid; j < n; j += kSpreadBT) {
        float4 p = make_float4(kFar, kFar, kFar, 0.f);
        if (j < M) p = in[g * stride + j];
        px[j] = p.x; py[j] = p.y; pz[j] = p.z;
        pm[j] = j < M ? 1.f : 0.f;
    }
    __syncthreads();
    const int q = q0 + (tid >> 2);
    const int qc = min(q, M - 1);
    const float ccx = px[qc], ccy = py[qc], ccz = pz[qc];
    const v2f cx = {ccx, ccx}, cy = {ccy, ccy}, cz = {ccz, ccz};
    const v2f kk = {k2, k2};
    float aw, ax, ay, az, am;
    int r;
    fit_pair_sums(px, py, pz, pm, n, sub, kk, cx, cy, cz, aw, ax, ay, az, am, r);
    float mv = 0.f;
    if (q < M && sub == 0) {
        const float nx = ax / aw, ny = ay / aw, nz = az / aw;
        const float ex = nx - ccx, ey = ny - ccy, ez = nz - ccz;
        mv = sqrtf(ex * ex + ey * ey + ez * ez);
        if (!(mv == mv)) mv = __uint_as_float(0x7f800000u);     // NaN never counts as converged
        // .w: the representative (< 4096) and the multiplicity (a count of ones, <= 4096) of the point, as bits
        out[g * stride + q] = make_float4(nx, ny, nz, __uint_as_float((unsigned)r | ((unsigned)am << 16)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mv = fmaxf(mv, __shfl_xor(mv, o));
    if ((tid & 63) == 0) atomicMax(&move_of[g], __float_as_uint(mv));     // moves are >= +0: their order is the order of their bits
}

// ---- the whole fit of one vote set in ONE workgroup (round 5) ---------------------------------------------------
// What the round-by-round kernels above cost on a batch of 8 frames x 5 objects (profiles/r05_pose_start.json): 903 launches, 80 ms,
// 2.5e11 weighted pairs -- the sets need 220-260 rounds to meet the reference's stopping rule (largest move < bandwidth / 1000), and every
// round is M^2 pairs.  But blurring mean shift COLLAPSES: after three rounds 1672 votes of a keypoint occupy 191 distinct fp32 positions,
// after eight rounds 68, and the remaining ~200 rounds move a few stragglers between those clusters.  Points at bit-identical positions
// move identically for ever (same sums in the same order), so they can be carried as ONE point with a multiplicity:
//     new_c_i = sum_u m_u w(c_i, c_u) c_u / sum_u m_u w(c_i, c_u)        over the distinct positions u
// is the reference's update (meanshift_pytorch.py:38-46) with equal terms collected -- the same iteration, not an approximation
// (the fp32 sums group differently: 1e-7-level differences, tests/test_pose_gpu.py keeps its 5e-4 m bar).
// One workgroup of 1024 threads owns a set of up to kCap points in LDS and makes ALL rounds without leaving the CU:
//   * pair loop as above (4 lanes per point, packed fp32, one v_exp_f32 per pair), weights multiplied by the multiplicity;
//   * the same loop notices d^2 == 0: the lowest index at a point's position is its representative, the multiplicities there add up;
//   * after a round with duplicates a block-wide scan compacts the representatives (order preserved) and re-targets the map
//     original point -> distinct position; the round's cost falls with the square of the collapse;
//   * stopping test, the ball count of the winner (multiplicity-weighted; ties to the lowest ORIGINAL index, :50-53), labels of the
//     original points and the centre are finished in the same launch: no host polling, no per-round launch.
// Sets larger than kCap points are left to the round-by-round path (ffb6d_mean_shift_f32 runs it for those sets only).
// Round 6: two sizes of the same kernel.  <1024, 4096> (the round-5 form) holds a CU for itself: 156 KB of LDS and 4 waves x 88 registers
// per SIMD -- next to the forward of the NEXT batch (ffb6d_amd/pipeline.py) no other workgroup fits beside it.  <512, 2048> (78 KB,
// 2 waves per SIMD) fits beside a GEMM or convolution workgroup, and two of them share a CU; it takes the sets of up to 2048 points
// (every object of a 5-object frame at N = 12288), the large form the sets above (m_lo = 2048).  Per point the sums run over the same
// lanes in the same order in both sizes: equal bits.
constexpr int kCap = 4096;                    // largest set the one-workgroup fit takes
constexpr int kCapLight = 2048;
constexpr int fit_lds_bytes(int cap) { return (3 * 2 + 2) * cap * (int)sizeof(float) + 3 * cap * (int)sizeof(unsigned short) + 512; }

template <int kBT, int kCap>
__global__ __launch_bounds__(kBT) void mean_shift_fit_kernel(
    const float4* __restrict__ sets, const int* __restrict__ counts, int sets_per_count, int64_t stride, float k2, float thresh,
    float bandwidth, int max_iter, float* __restrict__ centers, unsigned char* __restrict__ labels, int* __restrict__ n_inside,
    int* __restrict__ iters, int* __restrict__ rounds_ws, int* __restrict__ n_large, int m_lo, int count_large,
    const float4* __restrict__ pre0, const float4* __restrict__ pre1, const unsigned* __restrict__ pre_move, int G,
    const float4* __restrict__ sets_odd, const int* __restrict__ start_of) {
#pragma clang fp contract(fast)
    static_assert(kCap == 4 * kBT, "the compaction gives every thread four consecutive slots");
    // start_of != nullptr: the launch CONTINUES fits begun chip-wide (big_round_kernel / big_compact_kernel below): set g = the distinct
    // positions left after start_of[g] & 0xffff rounds, multiplicity in .w, in `sets` (even round count) or `sets_odd`; bit 30 = converged
    // already (no further round); counts = distinct positions per set, 0 = not one of these sets (nothing is written)
    const bool resumed = start_of != nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char fit_lds[];
    float* px = reinterpret_cast<float*>(fit_lds);              // [2][kCap] each: positions of the distinct points, ping-pong
    float* py = px + 2 * kCap;
    float* pz = py + 2 * kCap;
    float* pm = pz + 2 * kCap;                                  // [kCap] multiplicity
    float* pmn = pm + kCap;                                     // [kCap] multiplicity after merging (valid at representatives)
    unsigned short* rep = reinterpret_cast<unsigned short*>(pmn + kCap);   // [kCap] lowest index at the same position
    unsigned short* nidx = rep + kCap;                          // [kCap] index after compaction
    unsigned short* owner = nidx + kCap;                        // [kCap] original point -> distinct position
    float* red = reinterpret_cast<float*>(owner + kCap);        // [16 + 16 + ...] block reductions
    unsigned long long* red64 = reinterpret_cast<unsigned long long*>(red + 32);

    const int g = blockIdx.x;
    const int M = counts[g / sets_per_count];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (M > kCap) {                                             // a larger form / the round-by-round path takes this set
        if (tid == 0 && count_large) atomicAdd(n_large, 1);
        return;
    }
    if (M <= m_lo && m_lo > 0) return;                          // fitted by the smaller form (empty sets included)
    if (M <= 0 && resumed) return;
    if (M <= 0) {
        if (tid == 0) {
            centers[3 * g] = centers[3 * g + 1] = centers[3 * g + 2] = 0.f;
            if (n_inside) n_inside[g] = 0;
            if (iters) iters[g] = 0;
            rounds_ws[g] = 0;
        }
        if (labels)
            for (int64_t j = tid; j < stride; j += kBT) labels[g * stride + j] = 0;
        return;
    }
    const int start = resumed ? start_of[g] : 0;
    const int t_first = start & 0xffff;
    const float4* src = ((resumed && (t_first & 1)) ? sets_odd : sets) + g * stride;
    for (int j = tid; j < kCap; j += kBT) {
        const float4 p = j < M ? src[j] : make_float4(kFar, kFar, kFar, 0.f);
        px[j] = p.x; py[j] = p.y; pz[j] = p.z;
        px[kCap + j] = kFar; py[kCap + j] = kFar; pz[kCap + j] = kFar;      // the other buffer: far points wherever a round does not write
        pm[j] = j < M ? (resumed ? p.w : 1.f) : 0.f;
        owner[j] = (unsigned short)j;
    }
    int U = M, cur = 0, made = t_first;
    bool nodup0 = false;
    const int sub = tid & 3;
    const v2f kk = {k2, k2};
    __syncthreads();

    for (int t = (start >> 30) ? max_iter + 1 : t_first; t <= max_iter; ++t) {      // `it > max_iter` stops after max_iter + 1 rounds (:47)
        const float *cx_ = px + cur * kCap, *cy_ = py + cur * kCap, *cz_ = pz + cur * kCap;
        float *nx_ = px + (cur ^ 1) * kCap, *ny_ = py + (cur ^ 1) * kCap, *nz_ = pz + (cur ^ 1) * kCap;
        const int n = (U + 7) & ~7;                             // (slots U .. n hold far points of multiplicity 0)
        float move = 0.f;
        int dups = 0;
        // rounds 0 and 1 of the larger sets were made by mean_shift_spread_round_kernel (round 1: valid when round 0 merged nothing)
        const float4* ready = (M >= kSpreadMin && !resumed) ? (t == 0 ? pre0 : (t == 1 && nodup0 ? pre1 : nullptr)) : nullptr;
        if (ready) {
            for (int q = tid; q < U; q += kBT) {
                const float4 a4 = ready[g * stride + q];
                const unsigned bits = __float_as_uint(a4.w);
                nx_[q] = a4.x; ny_[q] = a4.y; nz_[q] = a4.z;
                pmn[q] = (float)(bits >> 16);
                rep[q] = (unsigned short)(bits & 0xffffu);
                dups += (int)(bits & 0xffffu) != q;
            }
            move = __uint_as_float(pre_move[t * G + g]);
        } else
        for (int i0 = 0; i0 < U; i0 += kBT / 4) {
            const int q = i0 + (tid >> 2);
            const int qc = min(q, U - 1);
            const float ccx = cx_[qc], ccy = cy_[qc], ccz = cz_[qc];
            const v2f cx = {ccx, ccx}, cy = {ccy, ccy}, cz = {ccz, ccz};
            float aw, ax, ay, az, am;
            int r;
            fit_pair_sums(cx_, cy_, cz_, pm, n, sub, kk, cx, cy, cz, aw, ax, ay, az, am, r);
            if (q < U && sub == 0) {
                const float nx = ax / aw, ny = ay / aw, nz = az / aw;
                nx_[q] = nx; ny_[q] = ny; nz_[q] = nz;
                pmn[q] = am;
                rep[q] = (unsigned short)r;
                dups += r != q;
                const float ex = nx - ccx, ey = ny - ccy, ez = nz - ccz;
                float mv = sqrtf(ex * ex + ey * ey + ez * ez);
                if (!(mv == mv)) mv = __uint_as_float(0x7f800000u);     // NaN never counts as converged
                move = fmaxf(move, mv);
            }
        }
        // block-wide: largest move, number of duplicates
        move = wave_max(move);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dups += __shfl_xor(dups, o);
        if (lane == 0) { red[wave] = move; red[16 + wave] = __int_as_float(dups); }
        __syncthreads();
        float bmove = red[0];
        int bdups = __float_as_int(red[16]);
        for (int i = 1; i < kBT / 64; ++i) { bmove = fmaxf(bmove, red[i]); bdups += __float_as_int(red[16 + i]); }
        made = t + 1;
        if (t == 0) nodup0 = bdups == 0;
        if (bdups == 0) {
            cur ^= 1;                                           // (slots U .. n of the other buffer: far points since the start or the last compaction)
        } else {
            // compact the representatives in index order: thread -> 4 consecutive slots
            int f[4], loc = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * tid + u;
                f[u] = i < U && rep[i] == i;
                loc += f[u];
            }
            int incl = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            __syncthreads();                                    // (red[] of the reductions above has been read by everyone)
            if (lane == 63) red[wave] = __int_as_float(incl);
            __syncthreads();
            int pre = 0, total = 0;
            for (int i = 0; i < kBT / 64; ++i) {
                const int c = __float_as_int(red[i]);
                pre += i < wave ? c : 0;
                total += c;
            }
            int k = pre + incl - loc;
            float *ox = px + cur * kCap, *oy = py + cur * kCap, *oz = pz + cur * kCap;      // the old positions are dead: compact into them
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * tid + u;
                if (i < U) nidx[i] = (unsigned short)k;         // (of a duplicate: overwritten below through its representative)
                if (f[u]) {
                    ox[k] = nx_[i]; oy[k] = ny_[i]; oz[k] = nz_[i];
                    pm[k] = pmn[i];                             // pm[k], k <= i: read only in the pair loop, which is over
                    ++k;
                }
            }
            __syncthreads();
            for (int j = tid; j < M; j += kBT) owner[j] = nidx[rep[owner[j]]];
            const int Un = total, nn = (Un + 7) & ~7;
            for (int j = Un + tid; j < max(nn, min(n, kCap)); j += kBT) {      // far points behind the new end, in both buffers
                ox[j] = kFar; oy[j] = kFar; oz[j] = kFar;
                nx_[j] = kFar; ny_[j] = kFar; nz_[j] = kFar;
                pm[j] = 0.f;
            }
            U = Un;
        }
        __syncthreads();
        if (bmove < thresh) break;
    }

    // ball sizes (multiplicity-weighted), winner = largest ball, ties to the lowest index = lowest original index (:50-53)
    const float *cx_ = px + cur * kCap, *cy_ = py + cur * kCap, *cz_ = pz + cur * kCap;
    const int n = (U + 7) & ~7;
    unsigned long long key = 0ull;
    for (int i0 = 0; i0 < U; i0 += kBT / 4) {
        const int q = i0 + (tid >> 2);
        const int qc = min(q, U - 1);
        const float ccx = cx_[qc], ccy = cy_[qc], ccz = cz_[qc];
        float inside = 0.f;
        for (int x = sub; x < n; x += 4) {
            const float dx = ccx - cx_[x], dy = ccy - cy_[x], dz = ccz - cz_[x];
            inside += sqrtf(dx * dx + dy * dy + dz * dz) < bandwidth ? pm[x] : 0.f;
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) inside += __shfl_xor(inside, o);
        if (q < U) {
            const unsigned long long mine = (static_cast<unsigned long long>((unsigned)inside) << 32) | (0xffffffffu - static_cast<unsigned>(q));
            key = mine > key ? mine : key;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if (lane == 0) red64[wave] = key;
    __syncthreads();
    for (int i = 0; i < kBT / 64; ++i) key = red64[i] > key ? red64[i] : key;
    const int win = static_cast<int>(0xffffffffu - static_cast<unsigned>(key & 0xffffffffull));
    const float wx = cx_[win], wy = cy_[win], wz = cz_[win];
    if (tid == 0) {
        centers[3 * g] = wx; centers[3 * g + 1] = wy; centers[3 * g + 2] = wz;
        if (n_inside) n_inside[g] = static_cast<int>(key >> 32);
        if (iters) iters[g] = made;
        rounds_ws[g] = made;
    }
    if (labels) {
        for (int64_t j = tid; j < stride; j += kBT) {
            unsigned char lab = 0;
            if (j < M) {
                const int u = owner[j];
                const float dx = wx - cx_[u], dy = wy - cy_[u], dz = wz - cz_[u];
                lab = sqrtf(dx * dx + dy * dy + dz * dz) < bandwidth;
            }
            labels[g * stride + j] = lab;
        }
    }
}

// ball size of every converged point, arg-max with ties to the lowest index (:50-53)
__global__ __launch_bounds__(kBlock) void ball_count_kernel(
    const float4* __restrict__ buf0, const float4* __restrict__ buf1, const int* __restrict__ counts,
    int sets_per_count, int64_t stride, float bandwidth, const int* __restrict__ rounds,
    unsigned long long* __restrict__ best, int min_cnt, const int* __restrict__ handed) {
    __shared__ float4 tile[kTile];
    __shared__ unsigned long long wbest[kBlock / 64];
    const int g = blockIdx.y;
    const int cnt = counts[g / sets_per_count];
    const int q0 = blockIdx.x * kPointsPerBlock;
    if (q0 >= cnt || cnt <= min_cnt || (handed && handed[g] <= min_cnt)) return;
    const float4* src = ((rounds[g] & 1) ? buf1 : buf0) + g * stride;
    const int sub = threadIdx.x & (kLanesPerPoint - 1);
    const int q = q0 + (threadIdx.x >> 2);
    const float4 c = src[min(q, cnt - 1)];
    int inside = 0;
    for (int j0 = 0; j0 < cnt; j0 += kTile) {
        __syncthreads();
        load_tile(tile, src, j0, cnt);
        __syncthreads();
        const int n = min(kTile, (cnt - j0 + 7) & ~7);
        for (int x = sub; x < n; x += kLanesPerPoint) {
            const float4 p = tile[x];
            const float dx = c.x - p.x, dy = c.y - p.y, dz = c.z - p.z;
            inside += sqrtf(dx * dx + dy * dy + dz * dz) < bandwidth;
        }
    }
#pragma unroll
    for (int o = 1; o < kLanesPerPoint; o <<= 1) inside += __shfl_xor(inside, o);
    unsigned long long key = 0ull;
    if (q < cnt) key = (static_cast<unsigned long long>(inside) << 32) | (0xffffffffu - static_cast<unsigned>(q));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kBlock / 64; ++i) key = wbest[i] > key ? wbest[i] : key;
        key = wbest[0] > key ? wbest[0] : key;
        atomicMax(&best[g], key);
    }
}

// centre = winning point, labels = membership of its ball (:54)
__global__ __launch_bounds__(kBlock) void ball_labels_kernel(
    const float4* __restrict__ buf0, const float4* __restrict__ buf1, const int* __restrict__ counts,
    int sets_per_count, int64_t stride, float bandwidth, const int* __restrict__ rounds,
    const unsigned long long* __restrict__ best, float* __restrict__ centers, unsigned char* __restrict__ labels,
    int* __restrict__ n_inside, int* __restrict__ iters, int min_cnt, const int* __restrict__ handed) {
    const int g = blockIdx.y;
    const int cnt = counts[g / sets_per_count];
    if (cnt <= min_cnt || (handed && handed[g] <= min_cnt)) return;     // results written by mean_shift_fit_kernel
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead && iters) iters[g] = rounds[g];
    if (cnt <= 0) {
        if (lead) {
            centers[3 * g] = centers[3 * g + 1] = centers[3 * g + 2] = 0.f;
            if (n_inside) n_inside[g] = 0;
        }
        if (labels && j < stride) labels[g * stride + j] = 0;
        return;
    }
    const float4* src = ((rounds[g] & 1) ? buf1 : buf0) + g * stride;
    const unsigned long long key = best[g];
    const int win = static_cast<int>(0xffffffffu - static_cast<unsigned>(key & 0xffffffffull));
    const float4 c = src[win];
    if (lead) {
        centers[3 * g] = c.x;
        centers[3 * g + 1] = c.y;
        centers[3 * g + 2] = c.z;
        if (n_inside) n_inside[g] = static_cast<int>(key >> 32);
    }
    if (labels && j < stride) {
        unsigned char lab = 0;
        if (j < cnt) {
            const float4 p = src[j];
            const float dx = c.x - p.x, dy = c.y - p.y, dz = c.z - p.z;
            lab = sqrtf(dx * dx + dy * dy + dz * dz) < bandwidth;
        }
        labels[g * stride + j] = lab;
    }
}

// ---- sets of more than kCap points: chip-wide rounds WITH duplicate merging, then the one-workgroup fit (round 6) -------------
// The round-by-round path above makes max_iter + 1 rounds of M^2 pairs: 200 ms for eight 12288-point sets.  But such a set collapses like
// the small ones -- 12288 scene points are 6246 distinct positions after 8 rounds, 3385 after 9, 63 after 22 -- so the work is in the first
// rounds only, IF equal positions are merged.  Here the rounds of the large sets run chip-wide on (position, multiplicity) lists:
//   big_round_kernel    64 points per workgroup against the whole list in 512-point tiles (the fit's pair function: weights times
//                       multiplicity, multiplicity and lowest index at the point's own position), new positions + representative
//   big_compact_kernel  one workgroup per set: representatives to the front in index order, original point -> distinct position map,
//                       the list length, the stopping rule (largest move < thresh after the round, :47)
// until every large set has at most kCap distinct positions (the host reads the lengths back after each round from the third on).
// mean_shift_fit_kernel then CONTINUES the fits from those lists (multiplicities as weights, the round counter where it stands) and
// big_labels_kernel carries its labels back to the original points.  A set that still has more than kCap positions after kBigRounds
// rounds, or converged with more, or has merged next to nothing after kBigGiveUp rounds (scattered points that never meet), is left
// to the round-by-round path from the start.
// Merging is exact (equal bits only): the same iteration as the reference's, terms of equal points collected (as in the fit).
constexpr int kBigRounds = 32;
constexpr int kBigGiveUp = 7;
constexpr int kCompactBT = 1024;

__global__ __launch_bounds__(kBlock) void big_init_kernel(
    const float4* __restrict__ sets, const int* __restrict__ counts, int sets_per_count, int64_t stride, const int* __restrict__ big_ids,
    float4* __restrict__ buf0, unsigned* __restrict__ owner, int* __restrict__ len_of) {
    const int g = big_ids[blockIdx.y];
    const int cnt = counts[g / sets_per_count];
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j == 0) len_of[g] = cnt;
    if (j < cnt) {
        float4 p = sets[g * stride + j];
        p.w = 1.f;
        buf0[g * stride + j] = p;
        owner[g * stride + j] = (unsigned)j;
    }
}

// state_of[g]: rounds made | converged << 30
__global__ __launch_bounds__(kBlock) void big_round_kernel(
    float4* __restrict__ buf0, float4* __restrict__ buf1, int64_t stride, const int* __restrict__ big_ids, float k2, int t,
    const int* __restrict__ len_of, const int* __restrict__ state_of, unsigned* __restrict__ rep, unsigned* __restrict__ move_of,
    int* __restrict__ dups_of) {
#pragma clang fp contract(fast)
    __shared__ __attribute__((aligned(16))) float tx[kTile], ty[kTile], tz[kTile], tm[kTile];
    const int g = big_ids[blockIdx.y];
    const int U = len_of[g];
    const int q0 = blockIdx.x * kPointsPerBlock;
    if ((state_of[g] >> 29) || q0 >= U) return;
    const float4* src = ((t & 1) ? buf1 : buf0) + g * stride;
    float4* dst = ((t & 1) ? buf0 : buf1) + g * stride;
    const int sub = threadIdx.x & 3;
    const int q = q0 + (threadIdx.x >> 2);
    const float4 c = src[min(q, U - 1)];
    const v2f cx = {c.x, c.x}, cy = {c.y, c.y}, cz = {c.z, c.z}, kk = {k2, k2};
    float aw = 0.f, ax = 0.f, ay = 0.f, az = 0.f, am = 0.f;
    int r = 1 << 30;
    for (int j0 = 0; j0 < U; j0 += kTile) {
        __syncthreads();
        for (int x = threadIdx.x; x < kTile; x += kBlock) {
            const int j = j0 + x;
            const float4 p = j < U ? src[j] : make_float4(kFar, kFar, kFar, 0.f);
            tx[x] = p.x; ty[x] = p.y; tz[x] = p.z; tm[x] = p.w;
        }
        __syncthreads();
        const int n = min(kTile, (U - j0 + 7) & ~7);
        float w_, x_, y_, z_, m_;
        int r_;
        fit_pair_sums(tx, ty, tz, tm, n, sub, kk, cx, cy, cz, w_, x_, y_, z_, m_, r_);
        aw += w_; ax += x_; ay += y_; az += z_; am += m_;
        r = min(r, j0 + r_);
    }
    float move = 0.f;
    int dup = 0;
    if (q < U && sub == 0) {
        const float nx = ax / aw, ny = ay / aw, nz = az / aw;
        dst[q] = make_float4(nx, ny, nz, am);                   // (.w of a representative: the merged multiplicity)
        rep[g * stride + q] = (unsigned)r;
        dup = r != q;
        const float ex = nx - c.x, ey = ny - c.y, ez = nz - c.z;
        move = sqrtf(ex * ex + ey * ey + ez * ez);
        if (!(move == move)) move = __uint_as_float(0x7f800000u);   // NaN never counts as converged
    }
    move = wave_max(move);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dup += __shfl_xor(dup, o);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&move_of[g], __float_as_uint(move));
        if (dup) atomicAdd(&dups_of[g], dup);
    }
}

__global__ __launch_bounds__(kCompactBT) void big_compact_kernel(
    float4* __restrict__ buf0, float4* __restrict__ buf1, int64_t stride, const int* __restrict__ big_ids, const int* __restrict__ counts,
    int sets_per_count, float thresh, int t, int* __restrict__ len_of, int* __restrict__ state_of, unsigned* __restrict__ move_of,
    int* __restrict__ dups_of, const unsigned* __restrict__ rep, unsigned* __restrict__ owner) {
    __shared__ int wsum[kCompactBT / 64];
    const int g = big_ids[blockIdx.x];
    if (state_of[g] >> 29) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int U = len_of[g], dups = dups_of[g];
    const float move = __uint_as_float(move_of[g]);
    float4* cur = ((t & 1) ? buf0 : buf1) + g * stride;         // what round t wrote
    float4* old = ((t & 1) ? buf1 : buf0) + g * stride;         // what it read: dead, its .w takes the new index of every representative
    const unsigned* rp = rep + g * stride;
    if (dups > 0) {
        int base = 0;
        for (int i0 = 0; i0 < U; i0 += 4 * kCompactBT) {
            int f[4], loc = 0;
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 4 * tid + u;
                f[u] = i < U && rp[i] == (unsigned)i;
                if (f[u]) v[u] = cur[i];
                loc += f[u];
            }
            int incl = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            __syncthreads();                                    // (wsum of the previous chunk has been read; every cur[i] of this chunk is in registers)
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int pre = 0, total = 0;
            for (int i = 0; i < kCompactBT / 64; ++i) {
                const int w = wsum[i];
                pre += i < wave ? w : 0;
                total += w;
            }
            int k = base + pre + incl - loc;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (f[u]) {
                    cur[k] = v[u];                              // k <= i: a slot of this chunk (already in registers) or of an earlier one
                    old[i0 + 4 * tid + u].w = __int_as_float(k);
                    ++k;
                }
            base += total;
        }
        __syncthreads();
        unsigned* own = owner + g * stride;
        const int M = counts[g / sets_per_count];
        for (int j = tid; j < M; j += kCompactBT) own[j] = (unsigned)__float_as_int(old[rp[own[j]]].w);
        if (tid == 0) len_of[g] = base;
    }
    if (tid == 0) {
        // given up (bit 29): after kBigGiveUp rounds nearly every position is still distinct -- scattered votes that will not meet
        // (a random-init network's); the set goes to the round-by-round path without spending the other chip-wide rounds
        const int len = len_of[g];
        const bool lost = t + 1 >= kBigGiveUp && len > kCap && len > counts[g / sets_per_count] - counts[g / sets_per_count] / 16;
        state_of[g] = (t + 1) | (move < thresh ? 1 << 30 : 0) | (lost ? 1 << 29 : 0);
        move_of[g] = 0u;
        dups_of[g] = 0;
    }
}

__global__ __launch_bounds__(kBlock) void big_labels_kernel(
    const int* __restrict__ big_ids, const int* __restrict__ counts, int sets_per_count, int64_t stride, const int* __restrict__ len_of,
    int cap, const unsigned char* __restrict__ fit_labels, const unsigned* __restrict__ owner, unsigned char* __restrict__ labels) {
    const int g = big_ids[blockIdx.y];
    if (len_of[g] > cap) return;                                // (left to the round-by-round path)
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= stride) return;
    labels[g * stride + j] = j < counts[g / sets_per_count] ? fit_labels[g * stride + owner[g * stride + j]] : (unsigned char)0;
}

template <typename MaskT>
__global__ __launch_bounds__(kBlock) void vote_sets_kernel(
    const float* __restrict__ pcld, const float* __restrict__ offsets, const MaskT* __restrict__ mask,
    const unsigned char* __restrict__ keep, const int* __restrict__ frame_of, const int* __restrict__ class_of,
    int S, int N, int64_t stride, float4* __restrict__ sets, int* __restrict__ counts) {
    __shared__ int wave_total[kBlock / 64];
    const int p = blockIdx.x;
    const int b = frame_of[p];
    const MaskT cls = static_cast<MaskT>(class_of[p]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int i0 = 0; i0 < N; i0 += kBlock) {
        const int i = i0 + threadIdx.x;
        bool sel = false;
        if (i < N) sel = mask[static_cast<int64_t>(b) * N + i] == cls && (!keep || keep[static_cast<int64_t>(b) * N + i]);
        const unsigned long long bal = __ballot(sel);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wave_total[wave] = __popcll(bal);
        __syncthreads();
        int pos = base + before;
        int total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            if (w < wave) pos += wave_total[w];
            total += wave_total[w];
        }
        if (sel) {
            const float* pt = pcld + (static_cast<int64_t>(b) * N + i) * 3;
            const float px = pt[0], py = pt[1], pz = pt[2];
            for (int s = 0; s < S; ++s) {
                const float* of = offsets + ((static_cast<int64_t>(b) * S + s) * N + i) * 3;
                sets[(static_cast<int64_t>(p) * S + s) * stride + pos] =
                    make_float4(px - of[0], py - of[1], pz - of[2], __int_as_float(i));
            }
        }
        base += total;
    }
    if (threadIdx.x == 0) counts[p] = base;
}

__global__ void labels_to_points_kernel(const float4* __restrict__ sets, const unsigned char* __restrict__ labels,
                                        const int* __restrict__ counts, const int* __restrict__ frame_of,
                                        int64_t stride, int N, unsigned char* __restrict__ keep) {
    const int p = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= counts[p]) return;
    const int i = __float_as_int(sets[p * stride + j].w);
    keep[static_cast<int64_t>(frame_of[p]) * N + i] = labels[p * stride + j];
}

template <typename MaskT>
__global__ void refine_mask_kernel(const float* __restrict__ pcld, const float* __restrict__ ctr_of,
                                   const MaskT* __restrict__ mask, const float* __restrict__ centers,
                                   const int* __restrict__ class_of, const int* __restrict__ pair_begin,
                                   const float* __restrict__ max_dist, int N, MaskT* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t at = static_cast<int64_t>(b) * N + i;
    MaskT m = mask[at];
    const int p0 = pair_begin[b], p1 = pair_begin[b + 1];
    if (m > 0 && p1 > p0) {
        const float vx = pcld[at * 3] - ctr_of[at * 3], vy = pcld[at * 3 + 1] - ctr_of[at * 3 + 1],
                    vz = pcld[at * 3 + 2] - ctr_of[at * 3 + 2];
        float dmin = __uint_as_float(0x7f800000u);
        int pmin = p0;
        for (int p = p0; p < p1; ++p) {
            const float dx = vx - centers[3 * p], dy = vy - centers[3 * p + 1], dz = vz - centers[3 * p + 2];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            if (d < dmin) {
                dmin = d;
                pmin = p;
            }
        }
        if (dmin < max_dist[pmin]) m = static_cast<MaskT>(class_of[pmin]);
    }
    out[at] = m;
}

// ---- 3x3 Kabsch in double, one thread per problem --------------------------------------------
__device__ void jacobi_eigen3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < 3; ++k) {      // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {      // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
}

__device__ inline void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ inline double normalize3(double* a) {
    const double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n > 0) {
        a[0] /= n;
        a[1] /= n;
        a[2] /= n;
    }
    return n;
}

// any unit vector orthogonal to u
__device__ inline void any_orthogonal(const double* u, double* o) {
    const int k = fabs(u[0]) <= fabs(u[1]) ? (fabs(u[0]) <= fabs(u[2]) ? 0 : 2) : (fabs(u[1]) <= fabs(u[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[k] = 1;
    cross3(u, e, o);
    normalize3(o);
}

__global__ void best_fit_kernel(const float* __restrict__ model, const float* __restrict__ found, int P, int n,
                                double* __restrict__ T) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* A = model + static_cast<int64_t>(p) * n * 3;
    const float* Bm = found + static_cast<int64_t>(p) * n * 3;
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            ca[k] += A[3 * i + k];
            cb[k] += Bm[3 * i + k];
        }
    for (int k = 0; k < 3; ++k) {
        ca[k] /= n;
        cb[k] /= n;
    }
    double H[3][3] = {};                       // H = AA^T BB (:49)
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[r][c] += (A[3 * i + r] - ca[r]) * (Bm[3 * i + c] - cb[c]);
    // H = U S V^T: eigenvectors of H^T H give V; u_k = H v_k / s_k
    double M[3][3], V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = H[0][r] * H[0][c] + H[1][r] * H[1][c] + H[2][r] * H[2][c];
    jacobi_eigen3(M, V);
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (M[order[j]][order[j]] > M[order[i]][order[i]]) {
                const int tmp = order[i];
                order[i] = order[j];
                order[j] = tmp;
            }
    double v[3][3], u[3][3];                   // rows = singular vectors, largest first
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) v[k][r] = V[r][order[k]];
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 3; ++r) u[k][r] = H[r][0] * v[k][0] + H[r][1] * v[k][1] + H[r][2] * v[k][2];
    const double s1 = normalize3(u[0]);
    if (!(s1 > 0)) {                           // H = 0: any rotation is optimal, return the identity
        u[0][0] = v[0][0] = 1; u[0][1] = u[0][2] = v[0][1] = v[0][2] = 0;
        u[1][1] = v[1][1] = 1; u[1][0] = u[1][2] = v[1][0] = v[1][2] = 0;
    } else {
        const double proj = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
        for (int r = 0; r < 3; ++r) u[1][r] -= proj * u[0][r];
        const double s2 = normalize3(u[1]);
        if (!(s2 > 1e-12 * s1)) any_orthogonal(u[0], u[1]);   // rank 1: the plane is free
    }
    // third pair by right-handedness on both sides == the reference's det(R) < 0 correction (:53-56):
    // R = V diag(1, 1, det(V U^T)) U^T does not depend on the sign choice of u3 / v3
    cross3(u[0], u[1], u[2]);
    cross3(v[0], v[1], v[2]);
    double R[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r][c] = v[0][r] * u[0][c] + v[1][r] * u[1][c] + v[2][r] * u[2][c];
    double* out = T + static_cast<int64_t>(p) * 12;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[4 * r + c] = R[r][c];
        out[4 * r + 3] = cb[r] - (R[r][0] * ca[0] + R[r][1] * ca[1] + R[r][2] * ca[2]);
    }
}

size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

}  // namespace

// A/B (ffb6d_pose_set_fit_form): 1 = sets of up to 2048 points on the light form of the one-workgroup fit (default), 0 = every set of
// up to 4096 points on the round-5 form
static int g_fit_form = 1;

extern "C" {

void ffb6d_pose_set_fit_form(int form) { g_fit_form = form; }
// rounds 0 and 1 of the sets of 512 .. 4096 points made chip-wide before the one-workgroup fits: 1 (default) = when the fits alone
// would leave at least half of the CUs idle (G <= CUs / 2: 40 centre-vote sets of a batch -- 2.21 -> 1.66 ms; with 320 keypoint
// sets the fits fill the chip and the spread rounds only add work, 3.17 -> 3.57 ms, profiles/r06_pose_spread_kernel_stats.txt),
// 2 = always, 0 = never
static int g_fit_spread = 1;
void ffb6d_pose_set_fit_spread(int on) { g_fit_spread = on; }
// sets of more than 4096 points: 1 (default) = chip-wide rounds with duplicate merging, then the one-workgroup fit; 0 = the round-by-round path
static int g_big_form = 1;
void ffb6d_pose_set_big_form(int form) { g_big_form = form; }


int ffb6d_vote_sets_f32(const float* pcld, const float* offsets, const void* mask, int mask_bits,
                        const unsigned char* keep, const int* frame_of, const int* class_of, int n_pairs, int B,
                        int S, int N, int64_t set_stride, float* sets, int* counts, ffb6d_stream_t stream) {
    FFB6D_REQUIRE(n_pairs >= 0 && B > 0 && S > 0 && N > 0, "vote_sets: bad sizes n_pairs=%d B=%d S=%d N=%d", n_pairs, B, S, N);
    FFB6D_REQUIRE(mask_bits == 32 || mask_bits == 64, "vote_sets: mask_bits must be 32 or 64, got %d", mask_bits);
    FFB6D_REQUIRE(set_stride >= N, "vote_sets: set_stride %lld < N %d", (long long)set_stride, N);
    if (n_pairs == 0) return 0;
    FFB6D_REQUIRE(pcld && offsets && mask && frame_of && class_of && sets && counts, "vote_sets: null pointer");
    hipStream_t st = ffb6d::as_stream(stream);
    if (mask_bits == 64)
        vote_sets_kernel<int64_t><<<n_pairs, kBlock, 0, st>>>(pcld, offsets, static_cast<const int64_t*>(mask), keep,
                                                               frame_of, class_of, S, N, set_stride,
                                                               reinterpret_cast<float4*>(sets), counts);
    else
        vote_sets_kernel<int32_t><<<n_pairs, kBlock, 0, st>>>(pcld, offsets, static_cast<const int32_t*>(mask), keep,
                                                               frame_of, class_of, S, N, set_stride,
                                                               reinterpret_cast<float4*>(sets), counts);
    FFB6D_LAUNCH_CHECK();
    return 0;
}

size_t ffb6d_mean_shift_workspace_bytes(int G, int64_t set_stride) {
    if (G <= 0 || set_stride <= 0) return 0;
    // two position buffers, the tail of the round-by-round path (shift, rounds, best), and for the chip-wide rounds of the sets beyond
    // the one-workgroup fit: representative and owner maps (u32 per point each) + five ints per set
    return 2 * align256(static_cast<size_t>(G) * set_stride * sizeof(float4)) + align256(3 * sizeof(unsigned) * G) +
           align256(sizeof(int) * G) + align256(sizeof(unsigned long long) * G) +
           2 * align256(static_cast<size_t>(G) * set_stride * sizeof(unsigned)) + align256(5 * sizeof(int) * G);
}

int ffb6d_mean_shift_f32(const float* sets, const int* counts, int sets_per_count, int G, int64_t set_stride,
                         int64_t max_count, float bandwidth, int max_iter, int check_every, float* centers,
                         unsigned char* labels,
                         int* n_inside, int* iters, void* workspace, size_t workspace_bytes, ffb6d_stream_t stream) {
    FFB6D_REQUIRE(G >= 0 && set_stride > 0 && sets_per_count > 0, "mean_shift: bad sizes G=%d stride=%lld", G,
                  (long long)set_stride);
    FFB6D_REQUIRE(bandwidth > 0.f && max_iter >= 0 && check_every >= 0, "mean_shift: bad bandwidth/max_iter");
    FFB6D_REQUIRE(max_count >= 0 && max_count <= set_stride, "mean_shift: max_count %lld outside [0, set_stride]",
                  (long long)max_count);
    if (G == 0) return 0;
    FFB6D_REQUIRE(sets && counts && centers, "mean_shift: null pointer");
    const size_t need = ffb6d_mean_shift_workspace_bytes(G, set_stride);
    if (!workspace || workspace_bytes < need)
        return ffb6d::set_error(FFB6D_ERR_WORKSPACE, "mean_shift: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = ffb6d::as_stream(stream);
    char* w = static_cast<char*>(workspace);
    const size_t buf_bytes = align256(static_cast<size_t>(G) * set_stride * sizeof(float4));
    float4* buf0 = reinterpret_cast<float4*>(w);
    float4* buf1 = reinterpret_cast<float4*>(w + buf_bytes);
    char* tail = w + 2 * buf_bytes;
    unsigned* shift = reinterpret_cast<unsigned*>(tail);
    int* rounds = reinterpret_cast<int*>(tail + align256(3 * sizeof(unsigned) * G));
    unsigned long long* best =
        reinterpret_cast<unsigned long long*>(tail + align256(3 * sizeof(unsigned) * G) + align256(sizeof(int) * G));
    const size_t tail_bytes = align256(3 * sizeof(unsigned) * G) + align256(sizeof(int) * G) + align256(sizeof(unsigned long long) * G);
    const size_t map_bytes = align256(static_cast<size_t>(G) * set_stride * sizeof(unsigned));
    unsigned* rep = reinterpret_cast<unsigned*>(tail + tail_bytes);
    unsigned* owner = reinterpret_cast<unsigned*>(tail + tail_bytes + map_bytes);
    int* big_ids = reinterpret_cast<int*>(tail + tail_bytes + 2 * map_bytes);
    int *len_of = big_ids + G, *state_of = big_ids + 2 * G, *dups_of = big_ids + 4 * G;
    unsigned* move_of = reinterpret_cast<unsigned*>(big_ids + 3 * G);
    FFB6D_HIP_TRY(hipMemsetAsync(tail, 0, tail_bytes, st));

    const double inv_bw2 = 1.0 / (static_cast<double>(bandwidth) * static_cast<double>(bandwidth));
    const float k2 = static_cast<float>(-0.5 * 1.4426950408889634 * inv_bw2);
    const float thresh = static_cast<float>(static_cast<double>(bandwidth) * 1e-3);   // stop_thresh (:30)

    // Sets of up to kCap points: the whole fit in one workgroup each, one launch for all of them (duplicate merging, no polling).
    // `best` doubles as the counter of the sets that are larger (one int at its start; cleared above, cleared again before the
    // round-by-round path uses the array).
    int* n_large = reinterpret_cast<int*>(best);
    bool fit_ok = true;                                        // a device that refuses the dynamic LDS: every set takes the round-by-round path
    {
        static int attr_set[ffb6d::kMaxDevices + 1];
        const int slot = ffb6d::device_slot();
        if (!ffb6d::cache_get(attr_set, slot)) {
            fit_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&mean_shift_fit_kernel<1024, kCap>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, fit_lds_bytes(kCap)) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&mean_shift_fit_kernel<512, kCapLight>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, fit_lds_bytes(kCapLight)) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&mean_shift_spread_round_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 4 * sizeof(float) * kCap) == hipSuccess;
            if (fit_ok) ffb6d::cache_set(attr_set, slot, 1);
            else (void)hipGetLastError();
        }
    }
    const int min_cnt = fit_ok ? kCap : -1;                    // sets above it are fitted round by round
    if (fit_ok) {
        const float4* s4 = reinterpret_cast<const float4*>(sets);
        // the first two rounds of the sets of kSpreadMin .. kCap points, 128 points per workgroup
        const float4 *pre0 = nullptr, *pre1 = nullptr;
        bool spread = g_fit_spread == 2;
        if (g_fit_spread == 1) {
            static int cu_count[ffb6d::kMaxDevices + 1];
            const int slot = ffb6d::device_slot();
            int cus = ffb6d::cache_get(cu_count, slot);
            if (cus == 0) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
                    cus = 256;
                ffb6d::cache_set(cu_count, slot, cus);
            }
            spread = 2 * G <= cus;
        }
        if (spread && set_stride >= kSpreadMin) {
            const int cap = static_cast<int>(std::min<int64_t>((set_stride + 7) & ~int64_t(7), kCap));
            const dim3 sgrid(static_cast<unsigned>(ceil_div(cap, kSpreadBT / 4)), static_cast<unsigned>(G));
            const size_t slds = 4 * sizeof(float) * static_cast<size_t>(cap);
            mean_shift_spread_round_kernel<<<sgrid, kSpreadBT, slds, st>>>(s4, counts, sets_per_count, set_stride, cap, k2, buf0, shift);
            FFB6D_LAUNCH_CHECK();
            mean_shift_spread_round_kernel<<<sgrid, kSpreadBT, slds, st>>>(buf0, counts, sets_per_count, set_stride, cap, k2, buf1, shift + G);
            FFB6D_LAUNCH_CHECK();
            pre0 = buf0;
            pre1 = buf1;
        }
        if (g_fit_form == 1) {
            mean_shift_fit_kernel<512, kCapLight><<<static_cast<unsigned>(G), 512, fit_lds_bytes(kCapLight), st>>>(
                s4, counts, sets_per_count, set_stride, k2, thresh, bandwidth, max_iter, centers, labels, n_inside, iters, rounds, n_large, 0, 0,
                pre0, pre1, shift, G, nullptr, nullptr);
            FFB6D_LAUNCH_CHECK();
            if (set_stride > kCapLight) {
                mean_shift_fit_kernel<1024, kCap><<<static_cast<unsigned>(G), 1024, fit_lds_bytes(kCap), st>>>(
                    s4, counts, sets_per_count, set_stride, k2, thresh, bandwidth, max_iter, centers, labels, n_inside, iters, rounds, n_large,
                    kCapLight, 1, pre0, pre1, shift, G, nullptr, nullptr);
                FFB6D_LAUNCH_CHECK();
            }
        } else {
            mean_shift_fit_kernel<1024, kCap><<<static_cast<unsigned>(G), 1024, fit_lds_bytes(kCap), st>>>(
                s4, counts, sets_per_count, set_stride, k2, thresh, bandwidth, max_iter, centers, labels, n_inside, iters, rounds, n_large, 0, 1,
                pre0, pre1, shift, G, nullptr, nullptr);
            FFB6D_LAUNCH_CHECK();
        }
    }
    if (fit_ok) {
        if (set_stride <= kCap) return 0;                      // no set can be larger
        int large = 0;                                         // one read-back per call (the old path polled every `check_every` rounds)
        FFB6D_HIP_TRY(hipMemcpyAsync(&large, n_large, sizeof(int), hipMemcpyDeviceToHost, st));
        FFB6D_HIP_TRY(hipStreamSynchronize(st));
        if (large == 0) return 0;
    }

    // blocks beyond a set's count exit at once: the largest count sizes the grids
    std::vector<int> host_counts(static_cast<size_t>(ceil_div(G, sets_per_count)));       // (the kernels index counts[g / sets_per_count], g < G)
    FFB6D_HIP_TRY(hipMemcpyAsync(host_counts.data(), counts, sizeof(int) * host_counts.size(), hipMemcpyDeviceToHost, st));
    FFB6D_HIP_TRY(hipStreamSynchronize(st));

    // ---- sets of more than kCap points: chip-wide rounds with duplicate merging, then the one-workgroup fit (round 6) ----
    const int* handed = nullptr;
    if (fit_ok && g_big_form) {
        std::vector<int> ids;
        int64_t span = 1;
        for (int g = 0; g < G; ++g) {
            const int c = host_counts[g / sets_per_count];
            if (c > kCap) { ids.push_back(g); span = std::max<int64_t>(span, std::min<int64_t>(c, set_stride)); }
        }
        const unsigned n_big = static_cast<unsigned>(ids.size());
        if (n_big == 0) return 0;                              // (every set was fitted above)
        FFB6D_HIP_TRY(hipMemsetAsync(big_ids, 0, 5 * sizeof(int) * G, st));
        FFB6D_HIP_TRY(hipMemcpyAsync(big_ids, ids.data(), sizeof(int) * n_big, hipMemcpyHostToDevice, st));
        big_init_kernel<<<dim3(static_cast<unsigned>(ceil_div(span, kBlock)), n_big), kBlock, 0, st>>>(reinterpret_cast<const float4*>(sets), counts, sets_per_count, set_stride,
                                                                                                     big_ids, buf0, owner, len_of);
        FFB6D_LAUNCH_CHECK();
        std::vector<int> host_state(2 * static_cast<size_t>(G));           // len_of | state_of: adjacent in the workspace, one copy
        bool polled = false;
        for (int t = 0; t <= max_iter && t < kBigRounds; ++t) {
            big_round_kernel<<<dim3(static_cast<unsigned>(ceil_div(span, kPointsPerBlock)), n_big), kBlock, 0, st>>>(
                buf0, buf1, set_stride, big_ids, k2, t, len_of, state_of, rep, move_of, dups_of);
            FFB6D_LAUNCH_CHECK();
            big_compact_kernel<<<n_big, kCompactBT, 0, st>>>(buf0, buf1, set_stride, big_ids, counts, sets_per_count, thresh, t, len_of, state_of,
                                                            move_of, dups_of, rep, owner);
            FFB6D_LAUNCH_CHECK();
            polled = false;
            if (t >= 2) {                                      // (nothing merges in the first rounds)
                FFB6D_HIP_TRY(hipMemcpyAsync(host_state.data(), len_of, 2 * sizeof(int) * G, hipMemcpyDeviceToHost, st));
                FFB6D_HIP_TRY(hipStreamSynchronize(st));
                polled = true;
                bool ready = true;
                span = 1;
                for (int g : ids) {
                    const bool conv = host_state[G + g] >> 29;          // converged or given up
                    ready = ready && (conv || host_state[g] <= kCap);
                    if (!conv) span = std::max<int64_t>(span, host_state[g]);
                }
                if (ready) break;
            }
        }
        if (!polled) {
            FFB6D_HIP_TRY(hipMemcpyAsync(host_state.data(), len_of, 2 * sizeof(int) * G, hipMemcpyDeviceToHost, st));
            FFB6D_HIP_TRY(hipStreamSynchronize(st));
        }
        // the fits continue from the lists (sets whose list is still longer: untouched, counted below)
        unsigned char* fit_labels = labels ? reinterpret_cast<unsigned char*>(rep) : nullptr;       // (the representatives are dead)
        if (g_fit_form == 1) {
            mean_shift_fit_kernel<512, kCapLight><<<static_cast<unsigned>(G), 512, fit_lds_bytes(kCapLight), st>>>(
                buf0, len_of, 1, set_stride, k2, thresh, bandwidth, max_iter, centers, fit_labels, n_inside, iters, rounds, n_large, 0, 0,
                nullptr, nullptr, nullptr, G, buf1, state_of);
            FFB6D_LAUNCH_CHECK();
        }
        mean_shift_fit_kernel<1024, kCap><<<static_cast<unsigned>(G), 1024, fit_lds_bytes(kCap), st>>>(
            buf0, len_of, 1, set_stride, k2, thresh, bandwidth, max_iter, centers, fit_labels, n_inside, iters, rounds, n_large,
            g_fit_form == 1 ? kCapLight : 0, 0, nullptr, nullptr, nullptr, G, buf1, state_of);
        FFB6D_LAUNCH_CHECK();
        if (labels) {
            big_labels_kernel<<<dim3(static_cast<unsigned>(ceil_div(set_stride, kBlock)), n_big), kBlock, 0, st>>>(
                big_ids, counts, sets_per_count, set_stride, len_of, kCap, fit_labels, owner, labels);
            FFB6D_LAUNCH_CHECK();
        }
        bool left = false;
        for (int g : ids) left = left || host_state[g] > kCap;
        if (!left) return 0;
        handed = len_of;                                       // the round-by-round kernels skip the sets with lists of <= kCap positions
    }

    // ---- what is left: one launch per round for all of those sets (round-1 path) ----
    FFB6D_HIP_TRY(hipMemsetAsync(best, 0, sizeof(unsigned long long) * G, st));
    FFB6D_HIP_TRY(hipMemsetAsync(shift, 0, 3 * sizeof(unsigned) * G, st));     // (the spread rounds of the fits used two of the slots)
    FFB6D_HIP_TRY(hipMemcpyAsync(buf0, sets, static_cast<size_t>(G) * set_stride * sizeof(float4),
                                 hipMemcpyDeviceToDevice, st));
    int64_t span = 1;
    for (int c : host_counts) span = std::max<int64_t>(span, std::min<int64_t>(c, set_stride));
    (void)max_count;
    const dim3 grid(static_cast<unsigned>(ceil_div(span, kPointsPerBlock)), static_cast<unsigned>(G));
    std::vector<float> host_shift;
    std::vector<int> host_handed;
    if (check_every > 0) host_shift.resize(G);
    if (handed) {
        host_handed.resize(G);
        FFB6D_HIP_TRY(hipMemcpyAsync(host_handed.data(), handed, sizeof(int) * G, hipMemcpyDeviceToHost, st));
        FFB6D_HIP_TRY(hipStreamSynchronize(st));
    }
    for (int t = 0; t <= max_iter; ++t) {       // `it > max_iter` stops after max_iter+1 rounds (:47)
        mean_shift_round_kernel<<<grid, kBlock, 0, st>>>(buf0, buf1, counts, sets_per_count, set_stride, k2, thresh,
                                                         shift, rounds, t, G, min_cnt, handed);
        FFB6D_LAUNCH_CHECK();
        if (check_every > 0 && (t + 1) % check_every == 0 && t < max_iter) {
            FFB6D_HIP_TRY(hipMemcpyAsync(host_shift.data(), shift + (t % 3) * G, sizeof(float) * G,
                                         hipMemcpyDeviceToHost, st));
            FFB6D_HIP_TRY(hipStreamSynchronize(st));
            bool all_done = true;
            for (int g = 0; g < G && all_done; ++g)
                all_done = host_counts[g / sets_per_count] <= min_cnt || (handed && host_handed[g] <= min_cnt) || host_shift[g] < thresh;
            if (all_done) break;
        }
    }
    ball_count_kernel<<<grid, kBlock, 0, st>>>(buf0, buf1, counts, sets_per_count, set_stride, bandwidth, rounds, best, min_cnt, handed);
    FFB6D_LAUNCH_CHECK();
    const dim3 lgrid(static_cast<unsigned>(labels ? ceil_div(set_stride, kBlock) : 1), static_cast<unsigned>(G));
    ball_labels_kernel<<<lgrid, kBlock, 0, st>>>(buf0, buf1, counts, sets_per_count, set_stride, bandwidth, rounds, best,
                                                 centers, labels, n_inside, iters, min_cnt, handed);
    FFB6D_LAUNCH_CHECK();
    return 0;
}

int ffb6d_set_labels_to_points(const float* sets, const unsigned char* labels, const int* counts, const int* frame_of,
                               int n_pairs, int64_t set_stride, int N, unsigned char* keep, ffb6d_stream_t stream) {
    FFB6D_REQUIRE(n_pairs >= 0 && set_stride > 0 && N > 0, "set_labels_to_points: bad sizes");
    if (n_pairs == 0) return 0;
    FFB6D_REQUIRE(sets && labels && counts && frame_of && keep, "set_labels_to_points: null pointer");
    const dim3 grid(static_cast<unsigned>(ceil_div(set_stride, kBlock)), static_cast<unsigned>(n_pairs));
    labels_to_points_kernel<<<grid, kBlock, 0, ffb6d::as_stream(stream)>>>(reinterpret_cast<const float4*>(sets), labels,
                                                                          counts, frame_of, set_stride, N, keep);
    FFB6D_LAUNCH_CHECK();
    return 0;
}

int ffb6d_refine_mask_by_center(const float* pcld, const float* ctr_offsets, const void* mask, int mask_bits,
                                const float* centers, const int* class_of, const int* pair_begin,
                                const float* max_dist, int B, int N, void* mask_out, ffb6d_stream_t stream) {
    FFB6D_REQUIRE(B > 0 && N > 0, "refine_mask: bad sizes B=%d N=%d", B, N);
    FFB6D_REQUIRE(mask_bits == 32 || mask_bits == 64, "refine_mask: mask_bits must be 32 or 64, got %d", mask_bits);
    FFB6D_REQUIRE(pcld && ctr_offsets && mask && pair_begin && mask_out, "refine_mask: null pointer");
    const dim3 grid(static_cast<unsigned>(ceil_div(N, kBlock)), static_cast<unsigned>(B));
    hipStream_t st = ffb6d::as_stream(stream);
    if (mask_bits == 64)
        refine_mask_kernel<int64_t><<<grid, kBlock, 0, st>>>(pcld, ctr_offsets, static_cast<const int64_t*>(mask), centers,
                                                              class_of, pair_begin, max_dist, N,
                                                              static_cast<int64_t*>(mask_out));
    else
        refine_mask_kernel<int32_t><<<grid, kBlock, 0, st>>>(pcld, ctr_offsets, static_cast<const int32_t*>(mask), centers,
                                                              class_of, pair_begin, max_dist, N,
                                                              static_cast<int32_t*>(mask_out));
    FFB6D_LAUNCH_CHECK();
    return 0;
}

int ffb6d_best_fit_transform_f32(const float* model, const float* found, int P, int n, double* T,
                                 ffb6d_stream_t stream) {
    FFB6D_REQUIRE(P >= 0 && n > 0, "best_fit_transform: bad sizes P=%d n=%d", P, n);
    if (P == 0) return 0;
    FFB6D_REQUIRE(model && found && T, "best_fit_transform: null pointer");
    best_fit_kernel<<<static_cast<unsigned>(ceil_div(P, 64)), 64, 0, ffb6d::as_stream(stream)>>>(model, found, P, n, T);
    FFB6D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
