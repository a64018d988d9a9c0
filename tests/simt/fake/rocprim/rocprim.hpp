// tests/simt/fake/rocprim/rocprim.hpp -- TEST INFRASTRUCTURE: the one rocPRIM entry point the product uses
// (rocprim::radix_sort_pairs on 64-bit keys) as a stable host sort, for the SIMT emulation build.
#pragma once
#include <algorithm>
#include <cstddef>
#include <numeric>
#include <vector>

#include <hip/hip_runtime.h>

namespace rocprim {

template <typename K, typename V>
hipError_t radix_sort_pairs(void* temp, size_t& temp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, size_t n,
                            unsigned begin_bit, unsigned end_bit, hipStream_t)
{
    if (!temp) { temp_bytes = 256; return hipSuccess; }
    const K mask = (end_bit >= 8 * sizeof(K) ? ~K(0) : ((K(1) << end_bit) - 1)) & ~((K(1) << begin_bit) - 1);
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), size_t(0));
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
    for (size_t i = 0; i < n; ++i) { keys_out[i] = keys_in[order[i]]; vals_out[i] = vals_in[order[i]]; }
    return hipSuccess;
}

}  // namespace rocprim
