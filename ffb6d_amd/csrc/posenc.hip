// ffb6d_amd/csrc/posenc.hip -- relative position encoding fused with the first shared MLP of the local feature
// aggregation (Building_block.forward, RandLANet.py:196-199, 216-223) for gfx950; see csrc/posenc_body.h.
// HBM-bound on its output: algorithmic bytes = 12 B N + idx + esz * 16 B N cout.
#include <algorithm>

#include "common.h"
#include "ffb6d_ops.h"
#include "posenc_body.h"

namespace ffb6d {
namespace {

constexpr int BLK = 256;

template <typename T, typename IdxT>
__global__ void __launch_bounds__(BLK)
posenc_mlp_pm_kernel(const posenc::MlpArgs a)
{
    posenc::mlp_body<T, IdxT>(a, (long long)blockIdx.x * BLK + threadIdx.x, (long long)gridDim.x * BLK);
}

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

extern "C" int ffb6d_posenc_mlp_pm(int dtype, const float* xyz, const void* idx, int idx_bits, const float* w, int64_t ldw,
                                   const float* bias, int act, void* out, int64_t B, int64_t N, int K, int64_t cout,
                                   ffb6d_stream_t stream)
{
    FFB6D_REQUIRE((dtype == 0 || dtype == 1) && (idx_bits == 32 || idx_bits == 64), "posenc_mlp_pm: dtype must be 0/1, idx_bits 32 or 64");
    const int VL = dtype == 1 ? 8 : 4;
    FFB6D_REQUIRE(B >= 0 && N >= 0 && K >= 1 && cout >= VL && cout % VL == 0 && cout <= 1024 && ldw >= 10,
                  "posenc_mlp_pm: bad shape (cout a multiple of %d, at most 1024; ldw >= 10)", VL);
    FFB6D_REQUIRE(act >= 0 && act <= 2, "posenc_mlp_pm: act must be 0 (none), 1 (relu) or 2 (leaky 0.2)");
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE((long long)B * N * K < (1LL << 31), "posenc_mlp_pm: more than 2^31 (point, neighbour) pairs in one launch");
    FFB6D_REQUIRE(xyz && idx && w && out && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "posenc_mlp_pm: null or unaligned pointer");
    posenc::MlpArgs a;
    a.xyz = xyz; a.idx = idx; a.w = w; a.bias = bias; a.out = out;
    a.N = (int)N; a.K = K; a.ldw = (int)ldw;
    a.q = (int)(cout / VL);
    a.slope = act == 0 ? 1.f : (act == 1 ? 0.f : 0.2f);
    a.pairs = (long long)B * N * K;
    // whole workgroups of threads, a multiple of q (256 % q == 0 for every q that divides 256; other q: lcm below), a few
    // thousand pairs per thread at the largest level so that the 10 * VL weight loads of a thread are amortised
    int64_t per_block = BLK;
    while (per_block % a.q) per_block += BLK;                  // threads per stride unit: multiple of 256 and of q
    const int64_t units = a.pairs * a.q;
    int64_t blocks = std::min<int64_t>(ceil_div(units, 4 * BLK), (int64_t)256 * 8);      // 4 pairs per thread and iteration
    blocks = std::max<int64_t>(ceil_div(blocks * BLK, per_block) * (per_block / BLK), per_block / BLK);
    const dim3 grid((unsigned)blocks);
    hipStream_t st = as_stream(stream);
    if (dtype == 1) {
        if (idx_bits == 64) hipLaunchKernelGGL((posenc_mlp_pm_kernel<__bf16, int64_t>), grid, dim3(BLK), 0, st, a);
        else hipLaunchKernelGGL((posenc_mlp_pm_kernel<__bf16, int32_t>), grid, dim3(BLK), 0, st, a);
    } else {
        if (idx_bits == 64) hipLaunchKernelGGL((posenc_mlp_pm_kernel<float, int64_t>), grid, dim3(BLK), 0, st, a);
        else hipLaunchKernelGGL((posenc_mlp_pm_kernel<float, int32_t>), grid, dim3(BLK), 0, st, a);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}
