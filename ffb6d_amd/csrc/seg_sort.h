// ffb6d_amd/csrc/seg_sort.h -- stable segmented radix sort of (32-bit key, 32-bit value) pairs, hand-written for gfx950 (csrc/seg_sort.hip).
// Used by the exact-KNN set preparation (Morton order of up to 8 point sets x B frames in ONE sort, csrc/knn_pruned.hip) and by the pose
// solver's duplicate merging (csrc/pose.hip).  Replaces rocprim::radix_sort_pairs on the hot path: for the ~1 M keys of an index
// pyramid rocPRIM ran its merge sort, 20 dependent launches = 0.13 ms alone and 0.40 ms inside the three-stream bench step
// (profiles/r04_rocprofv3_kernel_stats_steady_state.txt); this sort is 2 launches per 8-bit digit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace ffb6d {
namespace segsort {

constexpr int MAX_GROUPS = 8;
constexpr int CHUNK = 2048;                    // keys per workgroup (4 waves x 8 rounds x 64 lanes)

// A group = B segments of S keys each; segment b of the group is keys[pos0 + b * S, pos0 + (b + 1) * S).  Every segment is sorted by
// itself (keys of different segments never mix), stable, on the low `key_bits` bits of the key (the bits above are carried along).
struct Group {
    long long pos0;
    int S;
};
struct Plan {
    Group g[MAX_GROUPS];
    int blk0[MAX_GROUPS + 1];                  // first workgroup of group i (blk0[ngroups] = all workgroups)
    int ngroups, B;
};

// workgroups of a plan (fills blk0); bytes of the histogram scratch the sort needs
int plan_blocks(Plan& p);
size_t temp_bytes(const Plan& p);

// Sorts: ceil(key_bits / 8) passes, ping-pong between (keys, vals) and (keys_alt, vals_alt).  On return *sorted_in_alt tells where
// the result is (odd number of passes: the alt buffers).  Asynchronous on `st`.  Returns a hipError_t.
hipError_t sort_pairs(Plan& p, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, int key_bits, void* temp,
                      size_t temp_size, hipStream_t st, bool* sorted_in_alt);

}  // namespace segsort
}  // namespace ffb6d
