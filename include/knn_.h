/*
 * include/knn_.h -- forwarding header with the NAME of the reference's native KNN header
 * (ffb6d/models/RandLA/utils/nearest_neighbors/knn_.h), so that the reference's Cython binding
 * knn.pyx (`cdef extern from "knn_.h"`, knn.pyx:7-30) compiles UNCHANGED against this library:
 * put this directory on the include path instead of the reference's and link libffb6d_amd.so instead
 * of knn_.cxx (tests/test_capi_cpu.py builds exactly that).
 * The six entry points are declared, with C linkage, in ffb6d_knn.h.
 */
#ifndef FFB6D_KNN_FORWARD_H_
#define FFB6D_KNN_FORWARD_H_
#include "ffb6d_knn.h"
#endif
