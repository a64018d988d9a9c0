// ffb6d_amd/csrc/knn_pick.hip -- cpp_knn_batch_distance_pick[_omp] on gfx950.
//
// Reference: ffb6d/models/RandLA/utils/nearest_neighbors/knn_.cxx:138-271 (declared knn_.h:21-27, bound
// by knn.pyx:24-30,110-148).  Per frame the reference draws `nqueries` query points one after the other:
//     candidates = points whose use counter equals `current_id` (if none: current_id = min(counter))
//     index      = candidates[mt_rand() % candidates.size()]          (std::mt19937 seeded with time(0))
//     ids        = exact K-NN of points[index] among the frame's points (ascending distance)
//     counter[ids[k]] += 1,  counter[index] += 100
// and returns the queries and their neighbourhoods.  A draw depends on the counters left by all earlier
// draws, so a frame is inherently sequential in `nqueries`; frames are independent.
//
// Here one workgroup of 1024 threads owns a frame and runs the whole draw loop on the device:
//   * thread t owns the contiguous point range [t*chunk, (t+1)*chunk): candidate counting and the
//     selection of the r-th candidate in ascending index order (the order of the reference's
//     `possible_ids` vector) are a block-wide exclusive scan over per-thread counts;
//   * the K-NN of the drawn point is K rounds of a block-wide arg-min over packed 64-bit keys
//     (float bits of d^2 << 32 | index): ascending distance, ties to the lowest index, distances
//     evaluated as ((dx*dx + dy*dy) + dz*dz) without FMA contraction like every KNN of this library;
//   * std::mt19937 is restated on the device (one lane draws).  The reference's non-OpenMP variant
//     shares ONE generator across the frames of a batch, frame b consuming draws [b*nqueries,
//     (b+1)*nqueries): workgroup b discards b*nqueries draws first, so a batch reproduces the
//     reference's sequence for the same seed.  (The OpenMP variant races on that generator and is not
//     reproducible even with a fixed seed; both names run the same launch here.)
// Seed: time(0) as upstream, or the environment variable FFB6D_KNN_PICK_SEED (tests).
#include <cstdlib>
#include <ctime>

#include "common.h"

namespace ffb6d {
namespace {

constexpr int PICK_BLK = 1024;
constexpr int PICK_WAVES = PICK_BLK / 64;

struct Mt19937 {            // std::mt19937 (32-bit Mersenne twister), state in LDS, driven by one lane
    uint32_t mt[624];
    int mti;
};

__device__ void mt_seed(Mt19937& g, uint32_t seed)
{
    g.mt[0] = seed;
    for (int i = 1; i < 624; ++i) g.mt[i] = 1812433253u * (g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) + (uint32_t)i;
    g.mti = 624;
}

__device__ uint32_t mt_next(Mt19937& g)
{
    if (g.mti >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g.mt[i] & 0x80000000u) | (g.mt[(i + 1) % 624] & 0x7fffffffu);
            g.mt[i] = g.mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g.mti = 0;
    }
    uint32_t y = g.mt[g.mti++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

__global__ void __launch_bounds__(PICK_BLK)
distance_pick_kernel(const float* __restrict__ pts_all, float* __restrict__ queries_all, long long* __restrict__ idx_all,
                     int* __restrict__ used_all, int npts, int nq, int K, uint32_t seed)
{
    __shared__ Mt19937 gen;
    __shared__ int s_wave[PICK_WAVES];          // per-wave totals of the candidate scan / minima
    __shared__ unsigned long long s_key[PICK_WAVES];
    __shared__ int s_index, s_min;
    __shared__ unsigned int s_draw;
    __shared__ unsigned long long s_last;

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pts = pts_all + (size_t)b * npts * 3;
    int* used = used_all + (size_t)b * npts;
    float* queries = queries_all + (size_t)b * nq * 3;
    long long* idx = idx_all + (size_t)b * nq * K;
    const int chunk = (npts + PICK_BLK - 1) / PICK_BLK;
    const int i0 = min(npts, tid * chunk), i1 = min(npts, i0 + chunk);

    for (int i = i0; i < i1; ++i) used[i] = 0;
    if (tid == 0) {
        mt_seed(gen, seed);
        for (long long s = 0; s < (long long)b * nq; ++s) (void)mt_next(gen);   // draws of the earlier frames
    }
    int cur = 0;                                  // `current_id` of the reference, kept across draws
    __syncthreads();

    for (int q = 0; q < nq; ++q) {
        // ---- candidates: points with used == cur, in ascending index order -------------------------
        int mine, before, total;
        for (;;) {
            mine = 0;
            for (int i = i0; i < i1; ++i) mine += used[i] == cur;
            int incl = mine;                       // inclusive scan inside the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            int wbase = 0;
            total = 0;
            for (int w = 0; w < PICK_WAVES; ++w) {
                const int v = s_wave[w];
                if (w < wave) wbase += v;
                total += v;
            }
            before = wbase + incl - mine;
            __syncthreads();
            if (total > 0) break;
            int m = 0x7fffffff;                    // no candidate: current_id = min(used)
            for (int i = i0; i < i1; ++i) m = min(m, used[i]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
            if (lane == 0) s_wave[wave] = m;
            __syncthreads();
            if (tid == 0) {
                int mm = s_wave[0];
                for (int w = 1; w < PICK_WAVES; ++w) mm = min(mm, s_wave[w]);
                s_min = mm;
            }
            __syncthreads();
            cur = s_min;
            __syncthreads();
        }
        if (tid == 0) s_draw = mt_next(gen);
        __syncthreads();
        const int r = (int)((unsigned long long)s_draw % (unsigned long long)total);
        if (r >= before && r < before + mine) {
            int left = r - before;
            for (int i = i0; i < i1; ++i)
                if (used[i] == cur && left-- == 0) { s_index = i; break; }
        }
        if (tid == 0) s_last = 0ull;
        __syncthreads();
        const int index = s_index;
        const float qx = pts[index * 3], qy = pts[index * 3 + 1], qz = pts[index * 3 + 2];
        if (tid < 3) queries[q * 3 + tid] = pts[index * 3 + tid];

        // ---- exact K-NN of the drawn point: K rounds of block-wide arg-min over (d^2, index) -------
        for (int k = 0; k < K; ++k) {
            const unsigned long long last = s_last;
            unsigned long long best = ~0ull;
            for (int i = i0; i < i1; ++i) {
                const float dx = __fsub_rn(qx, pts[i * 3]), dy = __fsub_rn(qy, pts[i * 3 + 1]),
                            dz = __fsub_rn(qz, pts[i * 3 + 2]);
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
                // keys are unique (index in the low word); round 0 must accept key 0 as well
                const bool fresh = k == 0 || key > last;
                if (fresh && key < best) best = key;
            }
            best = wave_min_u64(best);
            if (lane == 0) s_key[wave] = best;
            __syncthreads();
            if (tid == 0) {
                unsigned long long m = s_key[0];
                for (int w = 1; w < PICK_WAVES; ++w) m = s_key[w] < m ? s_key[w] : m;
                s_last = m;
                const int id = (int)(unsigned)(m & 0xffffffffull);
                idx[(size_t)q * K + k] = id;
                used[id] += 1;
            }
            __syncthreads();
        }
        if (tid == 0) used[index] += 100;
        __threadfence_block();
        __syncthreads();
    }
}

void pick_host(const char* who, const float* pts, size_t B, size_t npts, size_t dim, float* queries, size_t nq, size_t K,
               long* out)
{
    if (B == 0 || nq == 0 || K == 0) return;
    int rc = FFB6D_OK;
    if (dim != 3) rc = set_error(FFB6D_ERR_ARG, "%s: dim must be 3 (got %zu)", who, dim);
    else if (K > 32 || npts < K) rc = set_error(FFB6D_ERR_ARG, "%s: need 1 <= K <= 32 and npts >= K (K=%zu, npts=%zu)", who, K, npts);
    else if (npts >= (1ull << 31) / 3 || nq >= (1ull << 31) / 32 || B >= 65536) rc = set_error(FFB6D_ERR_ARG, "%s: size too large", who);
    float *d_p = nullptr, *d_q = nullptr;
    long long* d_i = nullptr;
    int* d_u = nullptr;
    auto fail = [&](const char* what, hipError_t e) {
        rc = set_error(FFB6D_ERR_HIP, "%s: %s failed: %s", who, what, hipGetErrorString(e));
    };
    if (rc == FFB6D_OK) {
        const char* env = getenv("FFB6D_KNN_PICK_SEED");
        const uint32_t seed = env ? (uint32_t)strtoull(env, nullptr, 10) : (uint32_t)time(nullptr);   // knn_.cxx:143
        const size_t pb = B * npts * 3 * sizeof(float), qb = B * nq * 3 * sizeof(float), ib = B * nq * K * sizeof(long long);
        hipError_t e;
        static_assert(sizeof(long) == sizeof(long long), "LP64 expected");
        if ((e = hipMalloc(&d_p, pb)) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMalloc(&d_q, qb)) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMalloc(&d_i, ib)) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMalloc(&d_u, B * npts * sizeof(int))) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMemcpy(d_p, pts, pb, hipMemcpyHostToDevice)) != hipSuccess) fail("H2D", e);
        else {
            hipLaunchKernelGGL(distance_pick_kernel, dim3((unsigned)B), dim3(PICK_BLK), 0, nullptr, d_p, d_q, d_i, d_u,
                               (int)npts, (int)nq, (int)K, seed);
            if ((e = hipGetLastError()) != hipSuccess) fail("launch", e);
            else if ((e = hipMemcpy(out, d_i, ib, hipMemcpyDeviceToHost)) != hipSuccess) fail("D2H", e);
            else if ((e = hipMemcpy(queries, d_q, qb, hipMemcpyDeviceToHost)) != hipSuccess) fail("D2H", e);
        }
    }
    if (d_p) (void)hipFree(d_p);
    if (d_q) (void)hipFree(d_q);
    if (d_i) (void)hipFree(d_i);
    if (d_u) (void)hipFree(d_u);
    if (rc != FFB6D_OK)   // the reference signature is void: be loud, leave the outputs untouched
        fprintf(stderr, "[ffb6d_amd] %s failed: %s\n", who, ffb6d_last_error());
}

}  // namespace
}  // namespace ffb6d

extern "C" {

void cpp_knn_batch_distance_pick(const float* batch_data, const size_t batch_size, const size_t npts, const size_t dim,
                                 float* queries, const size_t nqueries, const size_t K, long* batch_indices)
{
    ffb6d::pick_host("cpp_knn_batch_distance_pick", batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices);
}

void cpp_knn_batch_distance_pick_omp(const float* batch_data, const size_t batch_size, const size_t npts, const size_t dim,
                                     float* batch_queries, const size_t nqueries, const size_t K, long* batch_indices)
{
    ffb6d::pick_host("cpp_knn_batch_distance_pick_omp", batch_data, batch_size, npts, dim, batch_queries, nqueries, K,
                     batch_indices);
}

}  // extern "C"
