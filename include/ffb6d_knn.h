/*
 * include/ffb6d_knn.h -- C ABI of the MI355X-native exact K-nearest-neighbour search.
 *
 * Drop-in for the reference's native KNN unit
 *     ffb6d/models/RandLA/utils/nearest_neighbors/knn_.h:4-27   (declarations)
 *     ffb6d/models/RandLA/utils/nearest_neighbors/knn_.cxx:22-135 (definitions)
 * which the reference binds from Cython (knn.pyx:7-30 `cdef extern from "knn_.h"`).
 * The six host-pointer entry points below keep the reference's names, argument order
 * and meaning (all six symbols knn.pyx declares, knn.pyx:7-30), exported with C linkage.
 * knn.pyx needs NO edit to bind this library: include/knn_.h is a forwarding header with the
 * reference header's name, so `cdef extern from "knn_.h"` picks these C-linkage declarations
 * up once <repo>/include is on the include path; tests/test_capi_cpu.py
 * compiles the reference's own knn.pyx that way and links it against libffb6d_amd.so.
 * The result is the exact K-NN set in ascending distance order;
 * squared distances are evaluated in float32 as ((dx*dx + dy*dy) + dz*dz) with no FMA
 * contraction (nanoflann.hpp:323-348); equal distances resolve to the LOWEST support
 * index (the reference's kd-tree keeps the first-visited member of a tie,
 * nanoflann.hpp:115-139 -- identical whenever distances are distinct).
 *
 * Differences from the reference, all loud instead of silent:
 *   - dim must be 3 and 1 <= K <= 32, npts >= K (the reference leaves stale output
 *     for npts < K, knn_.cxx:121-131); violations are reported through
 *     ffb6d_last_error() and the output is left untouched.
 *   - `_omp` and non-`_omp` variants are the same GPU launch (there is no host loop).
 *   - cpp_knn_batch_distance_pick[_omp] (knn_.cxx:138-271; unused by FFB6D) seed their
 *     std::mt19937 from time(0) like upstream unless FFB6D_KNN_PICK_SEED is set in the
 *     environment (tests); the _omp variant, which races on one generator upstream, returns
 *     the same result as the serial variant here.
 *
 * All pointers are plain host pointers for the cpp_* functions (the library stages them
 * through device memory itself) and plain device pointers for the ffb6d_*_device
 * functions.  No torch types cross this boundary.
 */
#ifndef FFB6D_KNN_H_
#define FFB6D_KNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t passed as an opaque pointer (NULL = the default stream). */
typedef void* ffb6d_stream_t;

/* 0 on success; negative on error (message via ffb6d_last_error()). */
#define FFB6D_OK 0
#define FFB6D_ERR_ARG (-1)
#define FFB6D_ERR_HIP (-2)
#define FFB6D_ERR_WORKSPACE (-3)

/* Thread-local description of the last error raised by any entry point. */
const char* ffb6d_last_error(void);

/* Library/ABI version (major*1000 + minor). */
int ffb6d_abi_version(void);

/* ---- host-pointer entry points: same names/signatures as knn_.h:4-19 ------------- */
/* points [npts,dim] f32, queries [nqueries,dim] f32, indices [nqueries,K] long (caller allocated) */
void cpp_knn(const float* points, const size_t npts, const size_t dim,
             const float* queries, const size_t nqueries,
             const size_t K, long* indices);
void cpp_knn_omp(const float* points, const size_t npts, const size_t dim,
                 const float* queries, const size_t nqueries,
                 const size_t K, long* indices);
/* batch_data [B,npts,dim], queries [B,nqueries,dim], batch_indices [B,nqueries,K] */
void cpp_knn_batch(const float* batch_data, const size_t batch_size, const size_t npts,
                   const size_t dim, const float* queries, const size_t nqueries,
                   const size_t K, long* batch_indices);
void cpp_knn_batch_omp(const float* batch_data, const size_t batch_size, const size_t npts,
                       const size_t dim, const float* queries, const size_t nqueries,
                       const size_t K, long* batch_indices);

/* knn_.h:21-27 / knn_.cxx:138-271: draws nqueries query points per frame (least-used points first, see
 * csrc/knn_pick.hip) and returns them with their K-NN.  batch_data [B,npts,dim] in; queries [B,nqueries,dim]
 * and batch_indices [B,nqueries,K] out (caller allocated, knn.pyx:133-134). */
void cpp_knn_batch_distance_pick(const float* batch_data, const size_t batch_size, const size_t npts,
                                 const size_t dim, float* queries, const size_t nqueries,
                                 const size_t K, long* batch_indices);
void cpp_knn_batch_distance_pick_omp(const float* batch_data, const size_t batch_size, const size_t npts,
                                     const size_t dim, float* batch_queries, const size_t nqueries,
                                     const size_t K, long* batch_indices);

/* ---- device-pointer entry points ------------------------------------------------ */
/* Scratch bytes ffb6d_knn_batch_device needs for this shape (0 is possible). */
size_t ffb6d_knn_workspace_bytes(int64_t batch_size, int64_t npts, int64_t nqueries, int K);

/*
 * support [B,npts,3] f32, query [B,nqueries,3] f32 (device, contiguous).
 * Any of the three outputs may be NULL:
 *   idx64 [B,nqueries,K] int64  -- what knn.pyx:93 allocates
 *   idx32 [B,nqueries,K] int32  -- what DataProcessing.knn_search returns (helper_tool.py:170)
 *   dist  [B,nqueries,K] f32    -- squared distances, ascending
 * workspace: device scratch of at least ffb6d_knn_workspace_bytes(...) bytes (may be NULL
 * when that is 0).  Stream-ordered on `stream`; does not synchronise.
 */
int ffb6d_knn_batch_device(const float* support, const float* query,
                           int64_t batch_size, int64_t npts, int64_t nqueries, int K,
                           int64_t* idx64, int32_t* idx32, float* dist,
                           void* workspace, size_t workspace_bytes, ffb6d_stream_t stream);

/* ---- prepared (spatially sorted) point sets ------------------------------------------------
 * The 22 searches of one FFB6D index pyramid reuse 8 point sets (4 cloud levels, 3 image
 * grids, ...).  A set is put into Morton order once (ffb6d_knn_prepare) and can then serve as
 * support and/or query of any number of ffb6d_knn_search_prepared calls; the search visits only
 * the 64-point tiles whose bounding box can still hold one of the K nearest neighbours.  Results
 * are identical to ffb6d_knn_batch_device (exact, same tie rule, original index space/order).
 *
 *   prepared   : opaque device buffer of ffb6d_knn_prepared_bytes(B, npts) bytes
 *   workspace  : device scratch of ffb6d_knn_prepare_workspace_bytes(B, npts) bytes (only needed
 *                during ffb6d_knn_prepare)
 */
size_t ffb6d_knn_prepared_bytes(int64_t batch_size, int64_t npts);
size_t ffb6d_knn_prepare_workspace_bytes(int64_t batch_size, int64_t npts);
int ffb6d_knn_prepare(const float* points /* [B,npts,3] device */, int64_t batch_size, int64_t npts,
                      void* prepared, size_t prepared_bytes, void* workspace, size_t workspace_bytes,
                      ffb6d_stream_t stream);

/* Several point sets prepared together (one Morton sort over their concatenation, every other pass with the set as one more
 * grid dimension): pts[i] [B,npts[i],3], prepared[i] of ffb6d_knn_prepared_bytes(B, npts[i]) bytes, i < nsets <= 8; the results
 * are byte-identical to nsets calls of ffb6d_knn_prepare. */
size_t ffb6d_knn_prepare_multi_workspace_bytes(int nsets, const int64_t* npts, int64_t batch_size);
int ffb6d_knn_prepare_multi(int nsets, const float* const* pts, const int64_t* npts, int64_t batch_size, void* const* prepared,
                            const size_t* prepared_bytes, void* workspace, size_t workspace_bytes, ffb6d_stream_t stream);
/* One search of a batch handed to ffb6d_knn_search_multi: the arguments of ffb6d_knn_search_prepared / ffb6d_knn_batch_device. */
typedef struct {
    const void* prep_support;   /* prepared set, or NULL */
    const void* prep_query;     /* prepared set, or NULL */
    const float* support;       /* raw [B,npts,3] (needed when the search is not routed to the Morton-ordered kernels) */
    const float* query;         /* raw [B,nqueries,3] */
    int64_t S, Q;               /* npts, nqueries */
    int K;
    int64_t* idx64;             /* outputs as ffb6d_knn_batch_device; any may be NULL */
    int32_t* idx32;
    float* dist;
} ffb6d_knn_search_t;

/* Several independent searches over the same B frames in as few launches as there are kernels involved (the 22 searches of
 * an index pyramid: 4).  Each search is routed like a single call would be -- ffb6d_knn_uses_pruning(B, S, Q, K): the
 * 16-lane row kernel for 2 <= K <= 16 (prepared support; prepared or raw queries), the K = 1 kernel (both sets prepared),
 * else the LDS-tiled scan on the raw arrays -- and gives the same results. */
int ffb6d_knn_search_multi(int nsearches, const ffb6d_knn_search_t* searches, int64_t batch_size, ffb6d_stream_t stream);

/* Queries: either a prepared set (prepared_query) or, for 2 <= K <= 16, the raw [B,nqueries,3]
 * device array (raw_query, prepared_query = NULL) -- small query sets need no preparation. */
int ffb6d_knn_search_prepared(const void* prepared_support, const void* prepared_query,
                              const float* raw_query, int64_t batch_size, int64_t npts, int64_t nqueries,
                              int K, int64_t* idx64, int32_t* idx32, float* dist, ffb6d_stream_t stream);

/* Instrumentation: while a device pointer to a zero-initialised 64-bit counter is set, every prepared-set search adds the
 * number of (query, support) pairs whose distance it actually evaluated; NULL (the default) switches the counting off.
 * Device-wide, not stream-ordered with respect to running kernels: set it between searches. */
int ffb6d_knn_set_pair_counter(unsigned long long* device_counter);

/* 1 when ffb6d_knn_batch_device would take the prepared/pruned route for this shape
 * (large support sets), 0 when it scans brute force. */
int ffb6d_knn_uses_pruning(int64_t batch_size, int64_t npts, int64_t nqueries, int K);

#ifdef __cplusplus
}
#endif
#endif /* FFB6D_KNN_H_ */
