// tests/hostsim/hostsim.hip -- TEST INFRASTRUCTURE: runs per-thread kernel bodies of the product library on the HOST.
//
// Kernels whose threads are independent (no cross-lane operations, no LDS, no barriers) keep their body in a
// `__host__ __device__` function under ffb6d_amd/csrc/*_body.h.  This file is compiled host-only
// (`hipcc --cuda-host-only`, tests/hostsim/build.py) and loops over the launch grid calling the same source the GPU runs,
// so that index arithmetic and numerics of a kernel can be checked against torch on a machine without a GPU.
// Nothing in the product path loads this library.
#include <cstdint>
#include <vector>

#include "posenc_body.h"
#include "upconv_body.h"

extern "C" int hostsim_upconv_combine(int dtype, const void* z, const float* shift, float slope, void* out, int64_t B, int64_t IH,
                                      int64_t IW, int64_t OH, int64_t OW, int64_t C, int blocked)
{
    using namespace ffb6d::upconv;
    const int VL = dtype == 1 ? 8 : 4;
    if (C % VL) return -1;
    CombineArgs a;
    a.z = z; a.shift = shift; a.out = out;
    a.IH = (int)IH; a.IW = (int)IW; a.OH = (int)OH; a.OW = (int)OW;
    a.q = (int)(C / VL);
    a.rh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;       // as the launcher in csrc/upconv.hip
    a.rw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    a.slope = slope;
    a.banded = 1; a.nbx = a.nby = 0;                                // launch order: not used by the bodies
    if (blocked == 4) {                                             // bf16 on half units (the launcher's bf16 rule, csrc/upconv.hip)
        if (dtype != 1 || OH != 2 * IH || OW != 2 * IW || OW % 4) return -2;
        a.q = (int)(C / 4);
        const int threads = (int)((OW / 4 * a.q + 255) / 256 * 256);
        for (int rb = 0; rb < (int)(B * OH / 2); ++rb)
            for (int t = 0; t < threads; ++t) combine_block_body<ffb6d::Bf16Half, 2>(a, rb, t);
        return 0;
    }
    if (blocked) {                                                  // the launcher's rule (csrc/upconv.hip)
        if (OH != 2 * IH || OW != 2 * IW || OW % 4) return -2;
        const int threads = (int)((OW / 4 * a.q + 255) / 256 * 256);
        for (int rb = 0; rb < (int)(B * OH / 2); ++rb)
            for (int t = 0; t < threads; ++t) {
                if (dtype == 1) {
                    if (blocked == 2) combine_block_body<__bf16, 2>(a, rb, t);
                    else combine_block_body<__bf16, 1>(a, rb, t);
                } else {
                    if (blocked == 2) combine_block_body<float, 2>(a, rb, t);
                    else combine_block_body<float, 1>(a, rb, t);
                }
            }
        return 0;
    }
    const int threads = (int)((OW * a.q + 255) / 256 * 256);        // whole workgroups: the surplus threads must bail out
    for (int row = 0; row < (int)(B * OH); ++row)
        for (int t = 0; t < threads; ++t) {
            if (dtype == 1) combine_body<__bf16>(a, row, t);
            else combine_body<float>(a, row, t);
        }
    return 0;
}

extern "C" int hostsim_posenc_mlp(int dtype, const float* xyz, const void* idx, int idx_bits, const float* w, int64_t ldw,
                                  const float* bias, int act, void* out, int64_t B, int64_t N, int K, int64_t cout, int64_t blocks)
{
    using namespace ffb6d::posenc;
    const int VL = dtype == 1 ? 8 : 4;
    if (cout % VL) return -1;
    MlpArgs a;
    a.xyz = xyz; a.idx = idx; a.w = w; a.bias = bias; a.out = out;
    a.N = (int)N; a.K = K; a.ldw = (int)ldw;
    a.q = (int)(cout / VL);
    a.slope = act == 0 ? 1.f : (act == 1 ? 0.f : 0.2f);
    a.pairs = (long long)B * N * K;
    const long long nthreads = blocks * 256;                    // the launcher's grid: a multiple of q
    if (nthreads % a.q) return -2;
    for (long long tid = 0; tid < nthreads; ++tid) {
        if (dtype == 1) { if (idx_bits == 64) mlp_body<__bf16, int64_t>(a, tid, nthreads); else mlp_body<__bf16, int32_t>(a, tid, nthreads); }
        else { if (idx_bits == 64) mlp_body<float, int64_t>(a, tid, nthreads); else mlp_body<float, int32_t>(a, tid, nthreads); }
    }
    return 0;
}

// share of (2 x 4 block, filter tap) pairs of one frame whose source positions follow the compile-time pattern, i.e. that
// combine_block_body<T, true> sends down tap_static (the same checks, on the launcher's scales)
extern "C" double hostsim_upconv_static_share(int64_t IH, int64_t IW)
{
    using namespace ffb6d::upconv;
    const int OH = (int)(2 * IH), OW = (int)(2 * IW);
    const float rh = (float)(IH - 1) / (float)(OH - 1), rw = (float)(IW - 1) / (float)(OW - 1);
    long long fast = 0, all = 0;
    for (int Y0 = 0; Y0 < OH; Y0 += 2)
        for (int X0 = 0; X0 < OW; X0 += 4)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    Axis2 ay;
                    Axis4 ax;
                    tap_axis<2>(ay, Y0, ky, OH, (int)IH, rh);
                    tap_axis<4>(ax, X0, kx, OW, (int)IW, rw);
                    bool ok = true;
                    for (int i = 0; i < 2; ++i) ok = ok && ay.a[i] == pattern_a(ky == 1, i) && ay.b[i] == ay.a[i] + 1;
                    for (int j = 0; j < 4; ++j) ok = ok && ax.a[j] == pattern_a(kx == 1, j) && ax.b[j] == ax.a[j] + 1;
                    fast += ok;
                    ++all;
                }
    return (double)fast / (double)all;
}

// the XCD-aware launch order (upconv::xcd_band_block): every logical workgroup exactly once over the padded 1-D grid, and the
// workgroups of one XCD (id % 8) form one contiguous row-major range.  0 = ok.
extern "C" int hostsim_xcd_band_check(unsigned nbx, unsigned nby)
{
    using namespace ffb6d::upconv;
    const unsigned nb = nbx * nby, grid = 8 * ((nb + 7) / 8);
    std::vector<int> seen(nb, 0);
    std::vector<long long> lo(8, -1), hi(8, -1), cnt(8, 0);
    for (unsigned id = 0; id < grid; ++id) {
        unsigned bx = ~0u, by = ~0u;
        if (!xcd_band_block(id, nbx, nby, bx, by)) continue;
        if (bx >= nbx || by >= nby) return 1;
        const long long l = (long long)by * nbx + bx;
        if (seen[l]++) return 2;
        const unsigned x = id & 7;
        if (lo[x] < 0 || l < lo[x]) lo[x] = l;
        if (l > hi[x]) hi[x] = l;
        ++cnt[x];
    }
    for (unsigned l = 0; l < nb; ++l) if (seen[l] != 1) return 3;
    for (int x = 0; x < 8; ++x) if (cnt[x] && hi[x] - lo[x] + 1 != cnt[x]) return 4;
    return 0;
}
