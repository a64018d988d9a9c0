// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Re-exports the REFERENCE's own KNN entry points (declared in the reference header
// ffb6d/models/RandLA/utils/nearest_neighbors/knn_.h:4-27, defined in knn_.cxx which is
// compiled in place from /root/reference by oracle/Makefile -- no reference source is
// copied into this repository) under unmangled names so ctypes can bind them.
#include <ctime>

#include "knn_.h"

// The reference seeds cpp_knn_batch_distance_pick with time(0) (knn_.cxx:143).  This library is linked
// with -Wl,-Bsymbolic-functions, so the reference object's call to `time` binds to the definition below:
// the reference code itself stays untouched and becomes reproducible for the pinning test.
static long g_fixed_time = -1;
extern "C" time_t time(time_t* t) noexcept
{
    time_t v;
    if (g_fixed_time >= 0) v = (time_t)g_fixed_time;
    else { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); v = ts.tv_sec; }
    if (t) *t = v;
    return v;
}

extern "C" {

void ref_cpp_knn(const float* points, size_t npts, size_t dim, const float* queries,
                 size_t nqueries, size_t K, long* indices)
{ cpp_knn(points, npts, dim, queries, nqueries, K, indices); }

void ref_cpp_knn_omp(const float* points, size_t npts, size_t dim, const float* queries,
                     size_t nqueries, size_t K, long* indices)
{ cpp_knn_omp(points, npts, dim, queries, nqueries, K, indices); }

void ref_cpp_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                       const float* queries, size_t nqueries, size_t K, long* batch_indices)
{ cpp_knn_batch(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices); }

void ref_cpp_knn_batch_omp(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                           const float* queries, size_t nqueries, size_t K, long* batch_indices)
{ cpp_knn_batch_omp(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices); }

void ref_cpp_knn_batch_distance_pick(const float* batch_data, size_t batch_size, size_t npts,
                                     size_t dim, float* queries, size_t nqueries, size_t K,
                                     long* batch_indices)
{ cpp_knn_batch_distance_pick(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices); }

void ref_set_fixed_time(long t) { g_fixed_time = t; }

}  // extern "C"
