// ffb6d_amd/csrc/train_rows.hip -- the neighbour operators of the TRAINING step on point-major rows, for gfx950.
//
// The reference trains under apex amp with its activations channel-major (train_lm.py:592-628; RandLANet.py:196-250,
// ffb6d.py:159-194); MIOpen's fast convolutions want channels-last.  Round 3's profile of the bf16 training step
// (profiles/r03_rocprofv3_kernel_stats_train_bf16.txt) had 15 ms of 82 in transposing copies and 5 ms in bf16 <-> fp32 casts
// around the channel-major fp32 neighbour operators.  The operators below read and write what the convolutions read and write:
// rows [points, C] of float32 or bfloat16 (= torch's channels_last memory of a [B,C,N,K] tensor), fp32 arithmetic inside.
//
//   forward gathers / max-pooling  : csrc/ops_pm.hip (ffb6d_gather_rows_pm, ffb6d_random_sample_pm) -- shared with inference
//   gather_sum_rows                : backward of every row gather (gather_neighbour RandLANet.py:225-234, nearest_interpolation
//                                    ffb6d.py:179-194, the `choose` pick ffb6d.py:309-312): grad[b, m, :] = sum of g[b,u,:] over the
//                                    u with idx[b,u] == m.  The caller inverts the index once (sort by destination -> CSR lists,
//                                    shared by all gathers through the same index tensor); a lane owns one unit of a DESTINATION
//                                    row, adds up the rows that reference it in fp32 and stores the activation type once.  (The
//                                    first version scattered with global float atomics: device-scope atomics leave the XCD's L2,
//                                    13.1 ms per step for the 19 gathers of the model; this form: 2.1 - 2.6 ms, profiles/r03_*rows*.)
//   random_sample_rows_bwd         : backward of FFB6D.random_sample (ffb6d.py:159-177): the gradient of an output element goes
//                                    to the neighbour that won the max -- recomputed from the rows (first maximum; a NaN wins like
//                                    in torch.max), not stored by the forward
//   att_pool_rows / _bwd           : softmax over the K neighbours, weighted sum (Att_pooling.forward, RandLANet.py:245-248) and
//                                    its gradient with respect to features and scores
//   log_softmax_rows / _bwd        : LogSoftmax over the channels of a map (`final` of the colour decoder, pspnet.py:108-112) with
//                                    the map's own element type in and out (fp32 arithmetic): under autocast ATen's version is
//                                    an fp32 operator -- the two full-resolution maps were cast up, written as fp32 and cast down
//                                    again by every consumer (~3 GB of traffic per step for 0.6 GB of data)
//
// A lane owns one 16-byte unit of a row (4 fp32 / 8 bf16 channels) so that a wave's loads and stores cover whole cache lines;
// sums are fp32.  Only the max-pool backward still uses float atomics (0.5 ms per step; its fp32 accumulator is rounded to the
// activation type by the caller).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "ffb6d_ops.h"
#include "row_unit.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

__device__ __forceinline__ bool wins(float v, float m) { return v > m || (v != v && m == m); }     // torch.max: NaN beats numbers

// out[r, :] = sum over j in [start[r], start[r+1]) of g[order[j], :];  thread = (destination row r, unit), four rows in flight
template <typename T>
__global__ void __launch_bounds__(BLK)
gather_sum_rows_kernel(const void* __restrict__ g, int ldq /* ldg / VL */, const int64_t* __restrict__ order,
                       const int64_t* __restrict__ start, void* __restrict__ out, int q, size_t total /* R*q */)
{
    using RU = RowUnit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t r = t / q;
    const int c = (int)(t - r * q);
    int64_t j = start[r];
    const int64_t j1 = start[r + 1];
    RU acc;
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) acc.v[e] = 0.f;
    for (; j + 4 <= j1; j += 4) {
        const int64_t u0 = order[j], u1 = order[j + 1], u2 = order[j + 2], u3 = order[j + 3];
        const RU a = RU::load(g, (size_t)u0 * ldq + c), b = RU::load(g, (size_t)u1 * ldq + c);
        const RU d = RU::load(g, (size_t)u2 * ldq + c), f = RU::load(g, (size_t)u3 * ldq + c);
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) acc.v[e] += (a.v[e] + b.v[e]) + (d.v[e] + f.v[e]);
    }
    for (; j < j1; ++j) {
        const RU a = RU::load(g, (size_t)order[j] * ldq + c);
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) acc.v[e] += a.v[e];
    }
    acc.store(out, t);
}

// The same sum with several lanes per (row, unit): the default wherever the units of a row are a power of two below 64 (the one-lane form
// above measured 111 us per call, 527 GB/s algorithmic: dependent-load latency at ~3 waves per SIMD; with this form and the row LogSoftmax
// the bf16 training step went from 61.2 to 54.0 ms, profiles/r04_start_bench_train_bf16_{default,optin}.json).
// A destination row has ~K readers (16 on average for the neighbour gathers, up to ~60) and only q = C / VL units: with one lane per
// unit the 16-byte loads of a row's readers would be issued one after the other by 2 .. 16 lanes.  Here L = 2^LOG_L lanes share a
// (row, unit): lane l adds readers l, l + L, ... (independent loads, all in flight), a butterfly over the L lanes finishes the sum.
// Lane order inside a group of L * q lanes: unit fastest (the q lanes that read one source row are adjacent: one contiguous
// segment), then l.  L * q <= 64, groups never straddle a wave; lanes past the end stay in the shuffles and skip the store.
template <typename T>
__global__ void __launch_bounds__(BLK)
gather_sum_rows_lanes_kernel(const void* __restrict__ g, int ldq /* ldg / VL */, const int64_t* __restrict__ order,
                             const int64_t* __restrict__ start, void* __restrict__ out, int q, int log_l, size_t total /* R * L * q */)
{
    using RU = RowUnit<T>;
    const int L = 1 << log_l;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    const bool live = t < total;
    const size_t tt = live ? t : total - 1;
    const size_t r = tt / ((size_t)L * q);
    const int in = (int)(tt - r * ((size_t)L * q));
    const int l = in / q, c = in - l * q;
    const int64_t j1 = start[r + 1];
    RU acc;
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) acc.v[e] = 0.f;
    int64_t j = start[r] + l;
    for (; j + L < j1; j += 2 * L) {                       // two readers per trip: both loads in flight
        const RU a = RU::load(g, (size_t)order[j] * ldq + c), b = RU::load(g, (size_t)order[j + L] * ldq + c);
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) acc.v[e] += a.v[e] + b.v[e];
    }
    if (j < j1) {
        const RU a = RU::load(g, (size_t)order[j] * ldq + c);
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) acc.v[e] += a.v[e];
    }
    for (int o = (q << log_l) >> 1; o >= q; o >>= 1)       // butterfly over l (lane distance q * 2^i); log_l = 0: no trip
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) acc.v[e] += __shfl_xor(acc.v[e], o, 64);
    if (live && l == 0) acc.store(out, r * q + c);
}

// thread = (output point pt = b*Np + n, unit): re-gathers the K rows, finds the winner per channel, adds g there
template <typename T, typename IdxT>
__global__ void __launch_bounds__(BLK)
random_sample_rows_bwd_kernel(const void* __restrict__ feat, const IdxT* __restrict__ idx, const void* __restrict__ g, int ldq,
                              float* __restrict__ acc, int q, int M, int Np, int K, size_t total /* B*Np*q */)
{
    using RU = RowUnit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;
    const int c = (int)(t - pt * q);
    const size_t b = pt / Np;
    const IdxT* ip = idx + pt * K;
    const size_t base = b * (size_t)M * q + c;
    RU best = RU::load(feat, base + (size_t)ip[0] * q);
    int arg[RU::VL];
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) arg[e] = 0;
    for (int k = 1; k < K; ++k) {
        const RU v = RU::load(feat, base + (size_t)ip[k] * q);
#pragma unroll
        for (int e = 0; e < RU::VL; ++e)
            if (wins(v.v[e], best.v[e])) { best.v[e] = v.v[e]; arg[e] = k; }
    }
    const RU gv = RU::load(g, pt * ldq + c);
#pragma unroll
    for (int e = 0; e < RU::VL; ++e)
        unsafeAtomicAdd(acc + (base + (size_t)ip[arg[e]] * q) * RU::VL + e, gv.v[e]);
}

// Attentive pooling on rows.  feat, scores: rows (pt*K + k) with strides ldf / lds; thread = (point, unit).
// KT > 0: the K = KT rows of both operands are fetched up front (2 KT independent 16-byte loads in flight per lane) and kept
// packed; KT = 0: any K, three passes that re-read the rows (L1 / L2 hits).
template <typename T, int KT> struct PoolRows {
    using RU = RowUnit<T>;
    const void *feat, *scores;
    size_t fbase, sbase;        // unit index of row k = 0
    int ldfq, ldsq, K;
    uint4 fr[KT > 0 ? KT : 1], sr[KT > 0 ? KT : 1];
    __device__ __forceinline__ void fetch()
    {
        if constexpr (KT > 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                sr[k] = static_cast<const uint4*>(scores)[sbase + (size_t)k * ldsq];
                fr[k] = static_cast<const uint4*>(feat)[fbase + (size_t)k * ldfq];
            }
        }
    }
    __device__ __forceinline__ RU f(int k) const
    {
        if constexpr (KT > 0) return RU::unpack(fr[k]);
        return RU::load(feat, fbase + (size_t)k * ldfq);
    }
    __device__ __forceinline__ RU s(int k) const
    {
        if constexpr (KT > 0) return RU::unpack(sr[k]);
        return RU::load(scores, sbase + (size_t)k * ldsq);
    }
    // softmax statistics and the pooled value: m = max_k s, den = sum_k exp(s - m), out = sum_k exp(s - m) f / den
    __device__ __forceinline__ void pool(float (&m)[RU::VL], float (&inv)[RU::VL], float (&out)[RU::VL]) const
    {
        const int n = KT > 0 ? KT : K;
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const RU v = s(k);
#pragma unroll
            for (int e = 0; e < RU::VL; ++e) m[e] = fmaxf(m[e], v.v[e]);
        }
        float den[RU::VL];
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) den[e] = 0.f, out[e] = 0.f;
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const RU v = s(k), x = f(k);
#pragma unroll
            for (int e = 0; e < RU::VL; ++e) {
                const float w = expf(v.v[e] - m[e]);
                den[e] += w;
                out[e] = fmaf(w, x.v[e], out[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) {
            inv[e] = 1.f / den[e];
            out[e] *= inv[e];
        }
    }
};

template <typename T, int KT>
__global__ void __launch_bounds__(BLK)
att_pool_rows_kernel(const void* __restrict__ feat, int ldfq, const void* __restrict__ scores, int ldsq, void* __restrict__ out, int q,
                     int K, size_t total /* P*q */)
{
    using RU = RowUnit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;
    const int c = (int)(t - pt * q);
    PoolRows<T, KT> pr;
    pr.feat = feat, pr.scores = scores, pr.ldfq = ldfq, pr.ldsq = ldsq, pr.K = K;
    pr.fbase = pt * K * ldfq + c, pr.sbase = pt * K * ldsq + c;
    pr.fetch();
    float m[RU::VL], inv[RU::VL];
    RU o;
    pr.pool(m, inv, o.v);
    o.store(out, t);
}

// gfeat[pt,k,:] = g * p_k ;  gscores[pt,k,:] = g * p_k * (f_k - out)      p = softmax_k(scores), out = sum_k p_k f_k
template <typename T, int KT>
__global__ void __launch_bounds__(BLK)
att_pool_rows_bwd_kernel(const void* __restrict__ g, int ldgq, const void* __restrict__ feat, int ldfq, const void* __restrict__ scores,
                         int ldsq, void* __restrict__ gfeat, void* __restrict__ gscores, int q, int K, size_t total /* P*q */)
{
    using RU = RowUnit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t pt = t / q;
    const int c = (int)(t - pt * q);
    PoolRows<T, KT> pr;
    pr.feat = feat, pr.scores = scores, pr.ldfq = ldfq, pr.ldsq = ldsq, pr.K = K;
    pr.fbase = pt * K * ldfq + c, pr.sbase = pt * K * ldsq + c;
    pr.fetch();
    const RU gv = RU::load(g, pt * ldgq + c);
    float m[RU::VL], inv[RU::VL], out[RU::VL];
    pr.pool(m, inv, out);
    const int n = KT > 0 ? KT : K;
    const size_t obase = pt * K * q + c;
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const RU v = pr.s(k), x = pr.f(k);
        RU gf, gs;
#pragma unroll
        for (int e = 0; e < RU::VL; ++e) {
            const float p = expf(v.v[e] - m[e]) * inv[e];
            gf.v[e] = gv.v[e] * p;
            gs.v[e] = gf.v[e] * (x.v[e] - out[e]);
        }
        gf.store(gfeat, obase + (size_t)k * q);
        gs.store(gscores, obase + (size_t)k * q);
    }
}

// y[r, :] = (x[r, :] - max) - log(sum exp(x[r, :] - max)): the q = C / VL lanes of a row (a power of two <= 64, consecutive lanes of
// one wave) reduce with butterfly shuffles; lanes past the end stay in the shuffles and skip the store.
template <typename T>
__global__ void __launch_bounds__(BLK)
log_softmax_rows_kernel(const void* __restrict__ x, void* __restrict__ y, int q, size_t total /* R*q */)
{
    using RU = RowUnit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    const bool live = t < total;
    const RU v = RU::load(x, live ? t : total - 1);
    float m = v.v[0];
#pragma unroll
    for (int e = 1; e < RU::VL; ++e) m = fmaxf(m, v.v[e]);
    for (int o = q >> 1; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) s += expf(v.v[e] - m);
    for (int o = q >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float ls = logf(s);
    RU o;
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) o.v[e] = (v.v[e] - m) - ls;
    if (live) o.store(y, t);
}

// gx[r, :] = g[r, :] - softmax(x[r, :]) * sum(g[r, :]); the softmax is recomputed from the forward's INPUT in fp32 (exp of the
// stored output would carry the output's bf16 rounding: 2^-9 |y| relative, percents for the unlikely channels)
template <typename T>
__global__ void __launch_bounds__(BLK)
log_softmax_rows_bwd_kernel(const void* __restrict__ g, const void* __restrict__ x, void* __restrict__ gx, int q, size_t total)
{
    using RU = RowUnit<T>;
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    const bool live = t < total;
    const size_t u = live ? t : total - 1;
    const RU gv = RU::load(g, u), xv = RU::load(x, u);
    float sg = 0.f, m = xv.v[0];
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) {
        sg += gv.v[e];
        m = fmaxf(m, xv.v[e]);
    }
    for (int o = q >> 1; o >= 1; o >>= 1) {
        sg += __shfl_xor(sg, o, 64);
        m = fmaxf(m, __shfl_xor(m, o, 64));
    }
    float p[RU::VL], s = 0.f;
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) {
        p[e] = expf(xv.v[e] - m);
        s += p[e];
    }
    for (int o = q >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float k = sg / s;
    RU o;
#pragma unroll
    for (int e = 0; e < RU::VL; ++e) o.v[e] = gv.v[e] - p[e] * k;
    if (live) o.store(gx, t);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool dt_ok(int dtype) { return dtype == 0 || dtype == 1; }
inline bool bits_ok(int bits) { return bits == 32 || bits == 64; }
inline unsigned blocks_for(size_t total) { return (unsigned)ceil_div((int64_t)total, BLK); }

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

#define FFB6D_ROWS_DT(dtype, T, ...)             \
    do {                                         \
        if (dtype == 1) { using T = __bf16; __VA_ARGS__ } else { using T = float; __VA_ARGS__ } \
    } while (0)
#define FFB6D_ROWS_IDX(bits, IdxT, ...)          \
    do {                                         \
        if (bits == 64) { using IdxT = int64_t; __VA_ARGS__ } else { using IdxT = int32_t; __VA_ARGS__ } \
    } while (0)

extern "C" int ffb6d_gather_sum_rows(int dtype, const void* g, int64_t ldg, const int64_t* order, const int64_t* start, void* out,
                                     int64_t R, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "gather_sum_rows: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(R >= 0 && C >= VL && C % VL == 0 && ldg >= C && ldg % VL == 0,
                  "gather_sum_rows: bad shape (C and ldg multiples of %d, ldg >= C)", VL);
    if (R == 0) return FFB6D_OK;
    FFB6D_REQUIRE(g && order && start && out && al16(g) && al16(out), "gather_sum_rows: null or unaligned pointer");
    FFB6D_REQUIRE(ldg / VL < (1LL << 31), "gather_sum_rows: too large");
    const int q = (int)(C / VL);
    // lanes per (row, unit): as many as fit a wave next to the q units, at most 8 (readers per row: ~16); needs q to be a power
    // of two (the butterfly's lane distances), else one lane per unit
    int log_l = 0;
    if ((q & (q - 1)) == 0)
        while (log_l < 3 && (q << (log_l + 1)) <= 64) ++log_l;
    if (log_l > 0) {
        const size_t total = ((size_t)R * q) << log_l;
        FFB6D_ROWS_DT(dtype, T, {
            hipLaunchKernelGGL((gather_sum_rows_lanes_kernel<T>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), g, (int)(ldg / VL),
                               order, start, out, q, log_l, total);
        });
    } else {
        const size_t total = (size_t)R * q;
        FFB6D_ROWS_DT(dtype, T, {
            hipLaunchKernelGGL((gather_sum_rows_kernel<T>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), g, (int)(ldg / VL), order,
                               start, out, q, total);
        });
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_random_sample_rows_bwd(int dtype, const void* feat, const void* idx, int idx_bits, const void* g, int64_t ldg,
                                            float* acc, int64_t B, int64_t M, int64_t C, int64_t Np, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype) && bits_ok(idx_bits), "random_sample_rows_bwd: dtype must be 0/1, idx_bits 32 or 64");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(B >= 0 && M >= 1 && Np >= 0 && K >= 1 && C >= VL && C % VL == 0 && ldg >= C && ldg % VL == 0,
                  "random_sample_rows_bwd: bad shape (C and ldg multiples of %d, ldg >= C)", VL);
    if (B == 0 || Np == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && g && acc && al16(feat) && al16(g) && al16(acc), "random_sample_rows_bwd: null or unaligned pointer");
    FFB6D_REQUIRE(ldg / VL < (1LL << 31) && M < (1LL << 31) && Np < (1LL << 31), "random_sample_rows_bwd: too large");
    const int q = (int)(C / VL);
    const size_t total = (size_t)B * Np * q;
    FFB6D_ROWS_DT(dtype, T, FFB6D_ROWS_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((random_sample_rows_bwd_kernel<T, IdxT>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), feat,
                           static_cast<const IdxT*>(idx), g, (int)(ldg / VL), acc, q, (int)M, (int)Np, K, total);
    }););
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_att_pool_rows(int dtype, const void* feat, int64_t ldf, const void* scores, int64_t lds, void* out, int64_t P,
                                   int K, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "att_pool_rows: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(P >= 0 && K >= 1 && C >= VL && C % VL == 0 && ldf >= C && lds >= C && ldf % VL == 0 && lds % VL == 0,
                  "att_pool_rows: bad shape (C and row strides multiples of %d)", VL);
    if (P == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && scores && out && al16(feat) && al16(scores) && al16(out), "att_pool_rows: null or unaligned pointer");
    FFB6D_REQUIRE(ldf / VL < (1LL << 31) && lds / VL < (1LL << 31), "att_pool_rows: too large");
    const int q = (int)(C / VL);
    const size_t total = (size_t)P * q;
    FFB6D_ROWS_DT(dtype, T, {
        if (K == 16)
            hipLaunchKernelGGL((att_pool_rows_kernel<T, 16>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), feat, (int)(ldf / VL),
                               scores, (int)(lds / VL), out, q, K, total);
        else
            hipLaunchKernelGGL((att_pool_rows_kernel<T, 0>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), feat, (int)(ldf / VL),
                               scores, (int)(lds / VL), out, q, K, total);
    });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_att_pool_rows_bwd(int dtype, const void* g, int64_t ldg, const void* feat, int64_t ldf, const void* scores,
                                       int64_t lds, void* gfeat, void* gscores, int64_t P, int K, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "att_pool_rows_bwd: dtype must be 0 (f32) or 1 (bf16)");
    const int VL = dtype ? 8 : 4;
    FFB6D_REQUIRE(P >= 0 && K >= 1 && C >= VL && C % VL == 0 && ldf >= C && lds >= C && ldg >= C && ldf % VL == 0 && lds % VL == 0 &&
                      ldg % VL == 0,
                  "att_pool_rows_bwd: bad shape (C and row strides multiples of %d)", VL);
    if (P == 0) return FFB6D_OK;
    FFB6D_REQUIRE(g && feat && scores && gfeat && gscores && al16(g) && al16(feat) && al16(scores) && al16(gfeat) && al16(gscores),
                  "att_pool_rows_bwd: null or unaligned pointer");
    FFB6D_REQUIRE(ldf / VL < (1LL << 31) && lds / VL < (1LL << 31) && ldg / VL < (1LL << 31), "att_pool_rows_bwd: too large");
    const int q = (int)(C / VL);
    const size_t total = (size_t)P * q;
    FFB6D_ROWS_DT(dtype, T, {
        if (K == 16)
            hipLaunchKernelGGL((att_pool_rows_bwd_kernel<T, 16>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), g, (int)(ldg / VL),
                               feat, (int)(ldf / VL), scores, (int)(lds / VL), gfeat, gscores, q, K, total);
        else
            hipLaunchKernelGGL((att_pool_rows_bwd_kernel<T, 0>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), g, (int)(ldg / VL),
                               feat, (int)(ldf / VL), scores, (int)(lds / VL), gfeat, gscores, q, K, total);
    });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

static bool lsm_shape_ok(int dtype, int64_t C)
{
    const int VL = dtype ? 8 : 4;
    if (C < VL || C % VL) return false;
    const int64_t q = C / VL;
    return q <= 64 && (q & (q - 1)) == 0;
}

extern "C" int ffb6d_log_softmax_rows(int dtype, const void* x, void* y, int64_t R, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "log_softmax_rows: dtype must be 0 (f32) or 1 (bf16)");
    FFB6D_REQUIRE(R >= 0 && lsm_shape_ok(dtype, C), "log_softmax_rows: C / %d must be a power of two <= 64", dtype ? 8 : 4);
    if (R == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && y && al16(x) && al16(y), "log_softmax_rows: null or unaligned pointer");
    const int q = (int)(C / (dtype ? 8 : 4));
    const size_t total = (size_t)R * q;
    FFB6D_ROWS_DT(dtype, T, { hipLaunchKernelGGL((log_softmax_rows_kernel<T>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), x, y, q, total); });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

extern "C" int ffb6d_log_softmax_rows_bwd(int dtype, const void* g, const void* x, void* gx, int64_t R, int64_t C, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(dt_ok(dtype), "log_softmax_rows_bwd: dtype must be 0 (f32) or 1 (bf16)");
    FFB6D_REQUIRE(R >= 0 && lsm_shape_ok(dtype, C), "log_softmax_rows_bwd: C / %d must be a power of two <= 64", dtype ? 8 : 4);
    if (R == 0) return FFB6D_OK;
    FFB6D_REQUIRE(g && x && gx && al16(g) && al16(x) && al16(gx), "log_softmax_rows_bwd: null or unaligned pointer");
    const int q = (int)(C / (dtype ? 8 : 4));
    const size_t total = (size_t)R * q;
    FFB6D_ROWS_DT(dtype, T, {
        hipLaunchKernelGGL((log_softmax_rows_bwd_kernel<T>), dim3(blocks_for(total)), dim3(BLK), 0, as_stream(stream), g, x, gx, q, total);
    });
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
