// ffb6d_amd/csrc/posenc_body.h -- per-thread body of the fused position encoding + first LFA shared MLP (csrc/posenc.hip).
//
// Reference: Building_block.forward, RandLANet.py:196-199:
//     f_xyz = relative_pos_encoding(xyz, neigh_idx)       [dis, p - q, p, q]  (10 channels, RandLANet.py:216-223)
//     f_xyz = mlp1(f_xyz)                                  Conv2d 1x1 10 -> d/2 + BatchNorm + activation
// With K = 10 the layer has 20 flop per output element and is bound by writing its output: the encoding is generated in
// registers and multiplied on the vector ALU, so the [B,N,16,10] tensor never exists (unfused: written padded to 16
// channels, read back by the GEMM).  A thread owns ONE 16-byte unit of the output row (4 fp32 / 8 bf16 channels), keeps
// the 10 weights of each of its channels in registers for its whole life and walks the (point, neighbour) pairs with a
// grid stride that is a multiple of the units per row; consecutive lanes store consecutive units: whole rows, coalesced.
// Encoding arithmetic as csrc/ops_pm.hip's rel_pos_enc_pm_kernel (separately rounded products and sums, IEEE sqrt);
// the dot product is a chain of fused multiply-adds in the order bias, dis, dx, dy, dz, p, q (GEMM bar 1e-5).
// __host__ __device__ and free of cross-lane operations: tests/hostsim runs it on the CPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "row_unit.h"

namespace ffb6d {
namespace posenc {

struct MlpArgs {
    const float* xyz;     // [B, N, 3]
    const void* idx;      // [B, N, K] int32 / int64 neighbour indices inside the frame
    const float* w;       // [cout, ldw] fp32, BatchNorm folded; columns 0..9 used
    const float* bias;    // [cout] fp32
    void* out;            // [B, N, K, cout] rows of T
    int N, K, ldw;
    int q;                // units per output row
    float slope;          // act(v) = max(v, slope * v): 1 = none, 0 = ReLU, 0.2 = LeakyReLU(0.2)
    long long pairs;      // B * N * K
};

template <typename T, typename IdxT>
__host__ __device__ __forceinline__ void mlp_body(const MlpArgs& a, long long tid, long long nthreads)
{
    using U = RowUnit<T>;
    const int u = (int)(tid % a.q);                       // nthreads % q == 0: the unit of a thread never changes
    const long long stride = nthreads / a.q;
    float w[U::VL][10], bias[U::VL];
#pragma unroll
    for (int e = 0; e < U::VL; ++e) {
        const float* row = a.w + (size_t)(u * U::VL + e) * a.ldw;
#pragma unroll
        for (int j = 0; j < 10; ++j) w[e][j] = row[j];
        bias[e] = a.bias ? a.bias[u * U::VL + e] : 0.f;
    }
    const IdxT* idx = static_cast<const IdxT*>(a.idx);
    // UN pairs per iteration: their index and coordinate loads are independent and in flight together (the chain
    // index -> neighbour coordinates -> store of one pair is two memory latencies long)
    constexpr int UN = 4;
    for (long long pr0 = tid / a.q; pr0 < a.pairs; pr0 += UN * stride) {
        float enc[UN][10];
        bool live[UN];
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            const long long pr = pr0 + n * stride;
            live[n] = pr < a.pairs;
            const unsigned pair = live[n] ? (unsigned)pr : 0u;     // pairs < 2^31 (launcher): 32-bit divisions
            const unsigned pn = pair / (unsigned)a.K;              // b * N + n
            const unsigned b = pn / (unsigned)a.N;
            const long long j = (long long)idx[pair];
            const float* p = a.xyz + (size_t)pn * 3;
            const float* qv = a.xyz + ((size_t)b * a.N + j) * 3;
            const float px = p[0], py = p[1], pz = p[2];
            const float qx = qv[0], qy = qv[1], qz = qv[2];
            const float dx = px - qx, dy = py - qy, dz = pz - qz;
            const float s = ((dx * dx) + (dy * dy)) + (dz * dz);      // compiled with -ffp-contract=off: no fusion
            const float row[10] = {sqrtf(s), dx, dy, dz, px, py, pz, qx, qy, qz};
#pragma unroll
            for (int t = 0; t < 10; ++t) enc[n][t] = row[t];
        }
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            if (!live[n]) continue;
            U o;
#pragma unroll
            for (int e = 0; e < U::VL; ++e) {
                float v = bias[e];
#pragma unroll
                for (int t = 0; t < 10; ++t) v = fmaf(w[e][t], enc[n][t], v);
                o.v[e] = fmaxf(v, a.slope * v);
            }
            o.store(a.out, (size_t)(pr0 + n * stride) * a.q + u);
        }
    }
}

}  // namespace posenc
}  // namespace ffb6d
