/*
 * oracle/knn_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by ffb6d_amd/).
 *
 * CPU restatement of the reference's exact K-nearest-neighbour search
 *   ffb6d/models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135  (cpp_knn_batch_omp)
 * as a plain float32 brute force.  The reference walks a nanoflann kd-tree; its
 * *result* is the exact K-NN set in ascending distance order, so the restatement
 * only has to honour the reference's arithmetic and ordering rules:
 *
 *   distance   nanoflann.hpp:323-348  L2_Adaptor::evalMetric, dim==3 takes the
 *              scalar tail loop: r = 0; r += d0*d0; r += d1*d1; r += d2*d2, with
 *              d = query[i] - point[i], every product and sum rounded to f32
 *              (x86-64 build without FMA => no contraction).
 *   result set nanoflann.hpp:79-145   KNNResultSet::addPoint: insertion sort that
 *              shifts only elements with dists[i-1] > dist (strict), so among
 *              equal distances the first-visited candidate stays in front.  A brute
 *              force scan in ascending index order therefore resolves ties to the
 *              LOWEST index; this is the tie rule of the whole project (the kd-tree
 *              traversal order is data dependent, so on exact ties the reference may
 *              pick another member of the tie -- tests compare distances there).
 *   output     long[B][Q][K], caller allocated (knn.pyx:93), row-major.
 *
 * Pinned against the real reference (oracle/_ref/libknn_ref.so built from the
 * reference's own knn_.cxx) in tests/test_oracle_cpu.py and through the hashes
 * in tests/golden/knn_pyramid_hashes.json.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <float.h>

static inline float sqdist3(const float *q, const float *p)
{
    /* keep every operation a separately rounded f32 op (file is built with
     * -ffp-contract=off; volatile-free but order-exact) */
    float d0 = q[0] - p[0];
    float d1 = q[1] - p[1];
    float d2 = q[2] - p[2];
    float r = 0.0f;
    r += d0 * d0;
    r += d1 * d1;
    r += d2 * d2;
    return r;
}

static inline float sqdist_n(const float *q, const float *p, size_t dim)
{
    /* nanoflann.hpp:330-346: groups of four are summed as
     * result += d0*d0 + d1*d1 + d2*d2 + d3*d3 (left-to-right), tail one by one */
    float r = 0.0f;
    size_t i = 0;
    while (i + 4 <= dim) {
        float d0 = q[i] - p[i], d1 = q[i + 1] - p[i + 1];
        float d2 = q[i + 2] - p[i + 2], d3 = q[i + 3] - p[i + 3];
        r += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        i += 4;
    }
    for (; i < dim; ++i) {
        float d = q[i] - p[i];
        r += d * d;
    }
    return r;
}

/* one query against one support set; ids/dists are K-long scratch */
static void knn_one(const float *pts, size_t npts, size_t dim, const float *q,
                    size_t K, int64_t *ids, float *dists)
{
    size_t count = 0;
    for (size_t j = 0; j < K; ++j) { dists[j] = FLT_MAX; ids[j] = 0; }
    for (size_t s = 0; s < npts; ++s) {
        float d = (dim == 3) ? sqdist3(q, pts + 3 * s) : sqdist_n(q, pts + dim * s, dim);
        /* nanoflann.hpp:1361: only candidates with d < worstDist reach addPoint */
        if (!(d < dists[K - 1])) continue;
        size_t i = count;
        for (; i > 0; --i) {
            if (dists[i - 1] > d) {
                if (i < K) { dists[i] = dists[i - 1]; ids[i] = ids[i - 1]; }
            } else break;
        }
        if (i < K) { dists[i] = d; ids[i] = (int64_t)s; }
        if (count < K) count++;
    }
}

/* same C signature as the reference's cpp_knn_batch_omp (knn_.h:16-19) */
void oracle_knn_batch(const float *batch_data, size_t batch_size, size_t npts, size_t dim,
                      const float *queries, size_t nqueries, size_t K, long *batch_indices)
{
    if (K == 0) return;
#pragma omp parallel
    {
        int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * K);
        float *dists = (float *)malloc(sizeof(float) * K);
#pragma omp for collapse(2) schedule(static)
        for (size_t b = 0; b < batch_size; ++b) {
            for (size_t i = 0; i < nqueries; ++i) {
                knn_one(batch_data + b * npts * dim, npts, dim,
                        queries + (b * nqueries + i) * dim, K, ids, dists);
                long *out = batch_indices + (b * nqueries + i) * K;
                for (size_t j = 0; j < K; ++j) out[j] = (long)ids[j];
            }
        }
        free(ids);
        free(dists);
    }
}

/* also return the squared distances (used by the duplicate-point tests, which
 * compare distance multisets instead of indices) */
void oracle_knn_batch_dist(const float *batch_data, size_t batch_size, size_t npts, size_t dim,
                           const float *queries, size_t nqueries, size_t K,
                           long *batch_indices, float *batch_dists)
{
    if (K == 0) return;
#pragma omp parallel
    {
        int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * K);
        float *dists = (float *)malloc(sizeof(float) * K);
#pragma omp for collapse(2) schedule(static)
        for (size_t b = 0; b < batch_size; ++b) {
            for (size_t i = 0; i < nqueries; ++i) {
                knn_one(batch_data + b * npts * dim, npts, dim,
                        queries + (b * nqueries + i) * dim, K, ids, dists);
                long *out = batch_indices + (b * nqueries + i) * K;
                float *od = batch_dists + (b * nqueries + i) * K;
                for (size_t j = 0; j < K; ++j) { out[j] = (long)ids[j]; od[j] = dists[j]; }
            }
        }
        free(ids);
        free(dists);
    }
}

/* ------------------------------------------------------------------------------------------
 * cpp_knn_batch_distance_pick (knn_.cxx:138-203): per frame `nqueries` sequential draws
 *   candidates = { i : used[i] == current_id }  (ascending i; if empty current_id = min(used))
 *   index      = candidates[mt_rand() % candidates.size()]
 *   ids        = K-NN of points[index];  used[ids[k]]++ ;  used[index] += 100
 * std::mt19937 (seeded with time(0) upstream) is restated below; ONE generator is shared by the
 * frames of a batch, consumed frame after frame.  Pinned against the reference's own function
 * compiled with a fixed clock (oracle/ref_shim.cpp, oracle/_ref/libknn_ref.so) in
 * tests/test_oracle_cpu.py. */
typedef struct { uint32_t mt[624]; int mti; } oracle_mt19937;

static void mt_seed(oracle_mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->mti = 624;
}

static uint32_t mt_next(oracle_mt19937 *g)
{
    if (g->mti >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->mti = 0;
    }
    uint32_t y = g->mt[g->mti++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void oracle_knn_batch_distance_pick(const float *batch_data, size_t batch_size, size_t npts, size_t dim,
                                    float *batch_queries, size_t nqueries, size_t K, long *batch_indices,
                                    uint32_t seed)
{
    if (K == 0 || npts == 0) return;
    oracle_mt19937 g;
    mt_seed(&g, seed);
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * K);
    float *dists = (float *)malloc(sizeof(float) * K);
    int *used = (int *)malloc(sizeof(int) * npts);
    size_t *cand = (size_t *)malloc(sizeof(size_t) * npts);
    for (size_t b = 0; b < batch_size; ++b) {
        const float *pts = batch_data + b * npts * dim;
        for (size_t i = 0; i < npts; ++i) used[i] = 0;
        int current_id = 0;
        for (size_t q = 0; q < nqueries; ++q) {
            size_t nc = 0;
            while (nc == 0) {
                for (size_t i = 0; i < npts; ++i)
                    if (used[i] == current_id) cand[nc++] = i;
                if (nc == 0) {
                    int m = used[0];
                    for (size_t i = 1; i < npts; ++i) if (used[i] < m) m = used[i];
                    current_id = m;
                }
            }
            size_t index = cand[(size_t)mt_next(&g) % nc];
            const float *query = pts + index * dim;
            knn_one(pts, npts, dim, query, K, ids, dists);
            for (size_t k = 0; k < K; ++k) used[ids[k]]++;
            used[index] += 100;
            for (size_t k = 0; k < K; ++k) batch_indices[(b * nqueries + q) * K + k] = (long)ids[k];
            for (size_t i = 0; i < dim; ++i) batch_queries[(b * nqueries + q) * dim + i] = query[i];
        }
    }
    free(ids); free(dists); free(used); free(cand);
}
