// ffb6d_amd/csrc/seg_sort.hip -- stable segmented LSD radix sort of (u32 key, u32 value) pairs for gfx950; see seg_sort.h.
//
// One 8-bit digit per pass, two launches per pass (three when a segment has more than 48 chunks: seg_prefix_kernel):
//   1. seg_hist_kernel:    a workgroup counts the digits of its chunk (2048 consecutive keys of one segment) -> hist[workgroup][256];
//   2. seg_scatter_kernel: thread d of a workgroup adds up the counts of digit d over the segment's workgroups (all of them: the digit's
//      total; those in front: what precedes this chunk) -- at most a few dozen 1 KB rows, L2 resident; a 256-wide exclusive scan of the
//      totals gives the digit's base.  Then every wave ranks ITS 512 keys (8 rounds of 64 consecutive keys): the lanes holding the same digit
//      find each other with 8 ballots, the lane's rank among them is a popcount, a per-wave LDS counter carries the digit's count from
//      round to round.  One barrier later the waves' counters are prefixed over the waves and every key goes to
//          segment start + base[digit] + in front of this chunk[digit] + in front of this wave[digit] + rank.
//      Keys keep their input order within a digit: stable, and deterministic (no atomics).
// Memory-wise a pass reads and writes 8 bytes per pair twice; for the ~1 M pairs of an index pyramid that is 30 MB per pass.
#include "seg_sort.h"

#include <algorithm>

namespace ffb6d {
namespace segsort {
namespace {

constexpr int BLK = 256;
constexpr int ROUNDS = CHUNK / BLK;            // 8 keys per thread

struct Where { int g, b, c, nblk; long long base; int S; };      // group, segment, chunk of the segment, chunks per segment, first key, keys

__device__ __forceinline__ Where locate(const Plan& p, int blk)
{
    int g = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUPS; ++i)
        if (i < p.ngroups && blk >= p.blk0[i]) g = i;
    Where w;
    w.g = g;
    w.S = p.g[g].S;
    w.nblk = (w.S + CHUNK - 1) / CHUNK;
    const int r = blk - p.blk0[g];
    w.b = r / w.nblk;
    w.c = r - w.b * w.nblk;
    w.base = p.g[g].pos0 + (long long)w.b * w.S;
    return w;
}

__global__ void __launch_bounds__(BLK)
seg_hist_kernel(const Plan p, const uint32_t* __restrict__ keys, int shift, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t cnt[256];
    const Where w = locate(p, blockIdx.x);
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* k = keys + w.base;
    const int i0 = w.c * CHUNK;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int i = i0 + r * BLK + threadIdx.x;
        if (i < w.S) atomicAdd(&cnt[(k[i] >> shift) & 255u], 1u);           // LDS integer atomics: the counts do not depend on the order
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + threadIdx.x] = cnt[threadIdx.x];
}

// Long segments (an image of 307 200 pixels is 150 chunks): the sums over a segment's chunks are made ONCE per segment instead of once
// per chunk -- hist[chunk][d] becomes the count of digit d in the chunks in front of it, totals[segment][d] the digit's total.
// One workgroup per segment, thread = digit, eight loads in flight.
__global__ void __launch_bounds__(BLK)
seg_prefix_kernel(const Plan p, uint32_t* __restrict__ hist, uint32_t* __restrict__ totals)
{
    int g = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUPS; ++i)
        if (i < p.ngroups && (int)blockIdx.x >= i * p.B) g = i;            // segments are numbered group-major: g * B + b
    const int b = blockIdx.x - g * p.B;
    const int nblk = (p.g[g].S + CHUNK - 1) / CHUNK;
    uint32_t* h = hist + ((size_t)p.blk0[g] + (size_t)b * nblk) * 256 + threadIdx.x;
    uint32_t run = 0;
    int c = 0;
    for (; c + 8 <= nblk; c += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = h[(size_t)(c + u) * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) { h[(size_t)(c + u) * 256] = run; run += v[u]; }
    }
    for (; c < nblk; ++c) { const uint32_t v = h[(size_t)c * 256]; h[(size_t)c * 256] = run; run += v; }
    totals[(size_t)blockIdx.x * 256 + threadIdx.x] = run;
}

template <bool PREFIXED>
__global__ void __launch_bounds__(BLK)
seg_scatter_kernel(const Plan p, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t* __restrict__ keys_out,
                   uint32_t* __restrict__ vals_out, int shift, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t off[256];              // first output slot (inside the segment) of this chunk's keys with digit d
    __shared__ uint32_t wcnt[4][256];          // per wave: keys with digit d seen so far; after the barrier: keys of the waves in front
    __shared__ uint32_t wsum[4];
    const Where w = locate(p, blockIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // digit d = threadIdx.x: total over the segment's chunks, and what the chunks in front of this one hold
        uint32_t total = 0, before = 0;
        if constexpr (PREFIXED) {              // seg_prefix_kernel has run
            before = hist[(size_t)blockIdx.x * 256 + threadIdx.x];
            total = totals[((size_t)w.g * p.B + w.b) * 256 + threadIdx.x];
        } else {
            const uint32_t* h = hist + (size_t)(blockIdx.x - w.c) * 256 + threadIdx.x;
            for (int c = 0; c < w.nblk; ++c) {
                const uint32_t v = h[(size_t)c * 256];
                total += v;
                before += c < w.c ? v : 0u;
            }
        }
        // exclusive scan of `total` over the 256 digits: inside the wave by shuffles, across the four waves through LDS
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
#pragma unroll
        for (int i = 0; i < 4; ++i) wcnt[i][threadIdx.x] = 0;
        __syncthreads();
        uint32_t pre = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) pre += i < wave ? wsum[i] : 0u;
        off[threadIdx.x] = pre + incl - total + before;
    }
    __syncthreads();

    const uint32_t* k = keys + w.base;
    const uint32_t* v = vals + w.base;
    const int i0 = w.c * CHUNK + wave * (CHUNK / 4);       // this wave's 512 consecutive keys
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t key[ROUNDS], val[ROUNDS], rank[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int i = i0 + r * 64 + lane;
        const bool live = i < w.S;
        key[r] = live ? k[i] : 0u;
        val[r] = live ? v[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool live = i0 + r * 64 + lane < w.S;
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long same = __ballot(live);           // live lanes holding the same digit as this lane
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long b = __ballot(one);
            same &= one ? b : ~b;
        }
        const uint32_t seen = wcnt[wave][d];                 // every lane of the group reads the count of the rounds before ...
        __builtin_amdgcn_sched_barrier(0);                   // (the reads of all lanes are issued before the write below)
        rank[r] = seen + (uint32_t)__popcll(same & lt);
        if (live && (same >> lane) <= 1ull) wcnt[wave][d] = seen + (uint32_t)__popcll(same);      // ... its last lane writes the new one
    }
    __syncthreads();
    {
        // digit d = threadIdx.x: counts of the four waves -> keys of the waves in front
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = wcnt[i][threadIdx.x];
            wcnt[i][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
    uint32_t* ko = keys_out + w.base;
    uint32_t* vo = vals_out + w.base;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (i0 + r * 64 + lane < w.S) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t dst = off[d] + wcnt[wave][d] + rank[r];
            ko[dst] = key[r];
            vo[dst] = val[r];
        }
    }
}

}  // namespace

int plan_blocks(Plan& p)
{
    int n = 0;
    for (int i = 0; i < p.ngroups; ++i) {
        p.blk0[i] = n;
        n += p.B * ((p.g[i].S + CHUNK - 1) / CHUNK);
    }
    for (int i = p.ngroups; i <= MAX_GROUPS; ++i) p.blk0[i] = n;
    return n;
}

size_t temp_bytes(const Plan& p)
{
    Plan q = p;
    return ((size_t)plan_blocks(q) + (size_t)p.ngroups * p.B) * 256 * sizeof(uint32_t);      // chunk histograms + per-segment totals
}

hipError_t sort_pairs(Plan& p, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, int key_bits, void* temp,
                      size_t temp_size, hipStream_t st, bool* sorted_in_alt)
{
    const int nblk = plan_blocks(p);
    *sorted_in_alt = false;
    if (nblk == 0 || key_bits <= 0) return hipSuccess;
    const int nseg = p.ngroups * p.B;
    if (temp_size < ((size_t)nblk + nseg) * 256 * sizeof(uint32_t)) return hipErrorInvalidValue;
    uint32_t* hist = static_cast<uint32_t*>(temp);
    uint32_t* totals = hist + (size_t)nblk * 256;
    int longest = 0;
    for (int i = 0; i < p.ngroups; ++i) longest = std::max(longest, (p.g[i].S + CHUNK - 1) / CHUNK);
    const bool prefixed = longest > 48;        // beyond a few dozen chunks per segment the per-chunk sums cost more than a launch
    bool alt = false;
    for (int shift = 0; shift < key_bits; shift += 8) {
        const uint32_t* ki = alt ? keys_alt : keys;
        const uint32_t* vi = alt ? vals_alt : vals;
        hipLaunchKernelGGL(seg_hist_kernel, dim3((unsigned)nblk), dim3(BLK), 0, st, p, ki, shift, hist);
        if (prefixed) {
            hipLaunchKernelGGL(seg_prefix_kernel, dim3((unsigned)nseg), dim3(BLK), 0, st, p, hist, totals);
            hipLaunchKernelGGL((seg_scatter_kernel<true>), dim3((unsigned)nblk), dim3(BLK), 0, st, p, ki, vi, alt ? keys : keys_alt,
                               alt ? vals : vals_alt, shift, hist, totals);
        } else {
            hipLaunchKernelGGL((seg_scatter_kernel<false>), dim3((unsigned)nblk), dim3(BLK), 0, st, p, ki, vi, alt ? keys : keys_alt,
                               alt ? vals : vals_alt, shift, hist, totals);
        }
        alt = !alt;
    }
    *sorted_in_alt = alt;
    return hipGetLastError();
}

}  // namespace segsort
}  // namespace ffb6d
