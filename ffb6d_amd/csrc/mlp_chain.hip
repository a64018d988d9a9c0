// ffb6d_amd/csrc/mlp_chain.hip -- three shared MLPs in a row as ONE launch for gfx950 (fp32): the layers after the first of a
// prediction head.
//
// Reference: FFB6D's heads (ffb6d/models/ffb6d.py:135-157: Conv1d(128,128) + BN + ReLU three times, then Conv1d(128, c); used at
// :316-318 on the [B, 128, N] fusion of the picked colour rows and the point rows).  forward_pm runs the first layers of the three
// heads as one stacked GEMM; what is left per head is 128 -> 128 -> 128 -> c on [B*N, 128] rows: as separate launches each layer is a
// K = 128 GEMM (four 32-byte k-steps: prologue and epilogue dominate, 0.35-0.45 of the fp32 MFMA peak) and the [B*N, 128] activations
// make two HBM round trips of 100 MB.
//
// Here a wave keeps its 32 points in registers through all three layers: with A = W (channels) and B = X (points), the accumulator
// register r of lane l holds D[channel (r&3) + 8*(r>>2) + 4*(l>>5)][point l&31] (csrc/mfma_pm.h) -- exactly the four consecutive k of the
// 16-byte chunk at 8*m + 4*(l>>5) (m = 4*tile + (r>>2)) that the same lane must supply as the B operand of the next layer's k-step m.
// So bias + activation are applied in place and the accumulators ARE the next operand: no LDS transpose, no store.  The weights
// (k-chunked on the host: [K/4][cout][4], ops_pm.k_chunked) are copied to LDS once per workgroup -- 144 KB, one workgroup per CU --
// and every workgroup walks its share of the 128-point tiles, the next tile's rows requested before the current tile is multiplied.
// Per output the products are summed in the k order of the tile kernels of csrc/mlp_pm.hip.
#include <algorithm>

#include "common.h"
#include "ffb6d_ops.h"
#include "mfma_pm.h"

namespace ffb6d {
namespace {

using namespace pm;

constexpr int D = 128;                 // width of the rows and of the two hidden layers

// bf16 (round 5; BASELINE configuration 5): v_mfma_f32_32x32x16_bf16 takes 8 consecutive k per lane, a k-step is 16 k.  The accumulator
// layout is the fp32 one (register r of lane l: channel (r&3) + 8*(r>>2) + 4*(l>>5) of point l&31), so a lane holds the channel groups
// {8g + 4kh + (0..3)} of every 32-channel tile and NOT eight consecutive channels.  A matrix product does not care in which order k is
// summed as long as both operands agree: k-step m of the next layer takes, from the lane with half kh, the groups g = 2(m&1) and
// 2(m&1)+1 of tile m>>1 -- channels base + (0..3) and base + 8 + (0..3), base = 32(m>>1) + 16(m&1) + 4kh -- and the host lays the
// weights of layers 2 and 3 out with their k-chunks permuted the same way (ops_pm.k_chunked(perm=True)).  The sixteen k of an MFMA are
// the same sixteen channels as in the separate launches, in another order inside the instruction.
template <typename T> struct Chain {
    static constexpr int SZ = El<T>::SZ;
    static constexpr int VL = 16 / SZ;                       // k per 16-byte chunk: 4 / 8
    static constexpr int NM = D / (2 * VL);                  // k-steps per layer: 16 / 8
    static constexpr int W12 = D * D;                        // elements of a k-chunked [128, 128] weight
    static constexpr int W3 = 32 * D;                        // ... of the last layer, its output channels padded to one 32-channel tile
    static constexpr size_t LDS = (size_t)(2 * W12 + W3) * SZ + (size_t)(2 * D + 32) * sizeof(float);
};

struct ChainParams {
    const void* x;                     // [rows, ldx] of T, the first 128 channels of a row are read
    const void *w1, *w2, *w3;          // k-chunked: [D/VL][128][VL], [D/VL][128][VL], [D/VL][32][VL] (rows >= cout3 of the last one are zeros)
    const float *b1, *b2, *b3;         // [128], [128], [cout3]
    float s1, s2, s3;                  // activation slopes (pm::activate: 0 = ReLU, 1 = none, 0.2 = LeakyReLU)
    void* out;                         // [rows, ldo] of T, channels [0, cout3) written
    int ldx, ldo, cout3, rows, n_tiles;
};

__device__ __forceinline__ void zero(f32x16& a)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// the B operand of k-step m from the previous layer's accumulator tiles: bias + activation, in the element type
template <typename T>
__device__ __forceinline__ u32x4 next_operand(const f32x16 (&acc)[4][1], const float* bias_lds, int m, int kh, float slope)
{
    if constexpr (El<T>::SZ == 4) {            // channels 8m + 4kh + (0..3): group m & 3 of tile m >> 2
        const f32x16& t = acc[m >> 2][0];
        const int g = m & 3;
        const float4 bb = *reinterpret_cast<const float4*>(bias_lds + 8 * m + 4 * kh);
        u32x4 b;
        b[0] = __float_as_uint(activate(t[4 * g + 0] + bb.x, slope));
        b[1] = __float_as_uint(activate(t[4 * g + 1] + bb.y, slope));
        b[2] = __float_as_uint(activate(t[4 * g + 2] + bb.z, slope));
        b[3] = __float_as_uint(activate(t[4 * g + 3] + bb.w, slope));
        return b;
    } else {                                   // groups 2(m&1), 2(m&1)+1 of tile m >> 1: channels base + (0..3), base + 8 + (0..3)
        const f32x16& t = acc[m >> 1][0];
        const int g = 2 * (m & 1), base = 32 * (m >> 1) + 16 * (m & 1) + 4 * kh;
        const float4 ba = *reinterpret_cast<const float4*>(bias_lds + base), bc = *reinterpret_cast<const float4*>(bias_lds + base + 8);
        const bf16x8 v = {(__bf16)activate(t[4 * g + 0] + ba.x, slope), (__bf16)activate(t[4 * g + 1] + ba.y, slope),
                          (__bf16)activate(t[4 * g + 2] + ba.z, slope), (__bf16)activate(t[4 * g + 3] + ba.w, slope),
                          (__bf16)activate(t[4 * g + 4] + bc.x, slope), (__bf16)activate(t[4 * g + 5] + bc.y, slope),
                          (__bf16)activate(t[4 * g + 6] + bc.z, slope), (__bf16)activate(t[4 * g + 7] + bc.w, slope)};
        return __builtin_bit_cast(u32x4, v);
    }
}

template <typename T>
__global__ void __launch_bounds__(BLK)
mlp_chain3_kernel(const ChainParams p)
{
    typedef Chain<T> C;
    constexpr int NM = C::NM, VL = C::VL;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [W1 | W2 | W3 | b1 | b2 | b3]
    T* w1s = reinterpret_cast<T*>(lds);
    T* w2s = w1s + C::W12;
    T* w3s = w2s + C::W12;
    float* b1s = reinterpret_cast<float*>(w3s + C::W3);
    float* b2s = b1s + D;
    float* b3s = b2s + D;
    for (int i = threadIdx.x; i < C::W12 / VL; i += BLK) {
        reinterpret_cast<u32x4*>(w1s)[i] = static_cast<const u32x4*>(p.w1)[i];
        reinterpret_cast<u32x4*>(w2s)[i] = static_cast<const u32x4*>(p.w2)[i];
    }
    for (int i = threadIdx.x; i < C::W3 / VL; i += BLK) reinterpret_cast<u32x4*>(w3s)[i] = static_cast<const u32x4*>(p.w3)[i];
    if (threadIdx.x < D) {
        b1s[threadIdx.x] = p.b1[threadIdx.x];
        b2s[threadIdx.x] = p.b2[threadIdx.x];
    }
    if (threadIdx.x < 32) b3s[threadIdx.x] = threadIdx.x < p.cout3 ? p.b3[threadIdx.x] : 0.f;     // cout3 entries are read, not 32
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kh = lane >> 5;
    int tile = blockIdx.x;
    if (tile >= p.n_tiles) return;
    const T* xb = static_cast<const T*>(p.x);
    u32x4 xc[NM], xn[NM];
    {
        const int r = min(tile * 128 + wave * 32 + l31, p.rows - 1);
        const T* row = xb + (size_t)r * p.ldx + VL * kh;
#pragma unroll
        for (int m = 0; m < NM; ++m) xc[m] = *reinterpret_cast<const u32x4*>(row + 2 * VL * m);
    }
    for (; tile < p.n_tiles; tile += gridDim.x) {
        {   // rows of this workgroup's next tile (the last iteration re-reads its own: no branch around the loads)
            const int nt = tile + (int)gridDim.x < p.n_tiles ? tile + (int)gridDim.x : tile;
            const int r = min(nt * 128 + wave * 32 + l31, p.rows - 1);
            const T* row = xb + (size_t)r * p.ldx + VL * kh;
#pragma unroll
            for (int m = 0; m < NM; ++m) xn[m] = *reinterpret_cast<const u32x4*>(row + 2 * VL * m);
        }
        f32x16 acc1[4][1], acc2[4][1], acc3[1][1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { zero(acc1[i][0]); zero(acc2[i][0]); }
        zero(acc3[0][0]);
        // The weight chunks of k-step m + 1 are requested from LDS before the MFMAs of k-step m; sched_barrier keeps the
        // compiler from hoisting all of a layer's operand registers to its top (which spills).
        u32x4 a[4], an[4];
        // layer 1: X from the registers loaded ahead
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4*>(w1s + (kh * D + 32 * i + l31) * VL);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const T* wn = m + 1 < NM ? w1s + (2 * (m + 1) + kh) * D * VL : w2s + kh * D * VL;       // ... or layer 2's first chunks
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4*>(wn + (32 * i + l31) * VL);
            u32x4 b[1] = {xc[m]};
            mfma_step<T, 4, 1>(acc1, a, b);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = an[i];
        }
        // layer 2: the accumulators of layer 1 are its operand
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m + 1 < NM) {
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const u32x4*>(w2s + ((2 * (m + 1) + kh) * D + 32 * i + l31) * VL);
            } else {
                an[0] = *reinterpret_cast<const u32x4*>(w3s + (kh * 32 + l31) * VL);                     // layer 3's first chunk
            }
            u32x4 b[1] = {next_operand<T>(acc1, b1s, m, kh, p.s1)};
            mfma_step<T, 4, 1>(acc2, a, b);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = an[i];
        }
        // layer 3: one 32-channel tile
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m + 1 < NM) an[0] = *reinterpret_cast<const u32x4*>(w3s + ((2 * (m + 1) + kh) * 32 + l31) * VL);
            u32x4 a3[1] = {a[0]}, b[1] = {next_operand<T>(acc2, b2s, m, kh, p.s2)};
            mfma_step<T, 1, 1>(acc3, a3, b);
            __builtin_amdgcn_sched_barrier(0);
            a[0] = an[0];
        }
        const int r = tile * 128 + wave * 32 + l31;
        if (r < p.rows) {
            T* orow = static_cast<T*>(p.out) + (size_t)r * p.ldo;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 8 * g + 4 * kh;
                if (c < p.cout3) {
                    const float4 bb = *reinterpret_cast<const float4*>(b3s + c);
                    float4 v;
                    v.x = activate(acc3[0][0][4 * g + 0] + bb.x, p.s3);
                    v.y = activate(acc3[0][0][4 * g + 1] + bb.y, p.s3);
                    v.z = activate(acc3[0][0][4 * g + 2] + bb.z, p.s3);
                    v.w = activate(acc3[0][0][4 * g + 3] + bb.w, p.s3);
                    El<T>::st4(orow + c, v);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) xc[m] = xn[m];
    }
}

float slope_of(int act) { return act == 1 ? 0.f : (act == 2 ? 0.2f : 1.f); }

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

namespace ffb6d {
namespace {

template <typename T>
int mlp_chain3_impl(const void* x, int64_t ldx, const void* w1k, const float* b1, int act1, const void* w2k, const float* b2, int act2,
                    const void* w3k, const float* b3, int act3, void* out, int64_t ldo, int64_t rows, int64_t cout3, ffb6d_stream_t stream)
{
    constexpr int SZ = El<T>::SZ, AL = 16 / SZ;
    FFB6D_REQUIRE(rows >= 0 && rows < (1LL << 31) - 256 && ldx >= D && ldx % AL == 0 && cout3 >= 4 && cout3 <= 32 && cout3 % 4 == 0 &&
                      ldo >= cout3 && ldo % 4 == 0,
                  "mlp_chain3_pm: rows of 128 channels (16-byte aligned row stride), 4 <= cout3 <= 32 a multiple of 4, ldo a multiple of 4");
    FFB6D_REQUIRE(ldx < (1LL << 31) && ldo < (1LL << 31), "mlp_chain3_pm: row strides must fit 31 bits (got %lld, %lld)", (long long)ldx,
                  (long long)ldo);
    FFB6D_REQUIRE(act1 >= 0 && act1 <= 2 && act2 >= 0 && act2 <= 2 && act3 >= 0 && act3 <= 2, "mlp_chain3_pm: act must be 0, 1 or 2");
    if (rows == 0) return FFB6D_OK;
    FFB6D_REQUIRE(x && w1k && b1 && w2k && b2 && w3k && b3 && out, "mlp_chain3_pm: null pointer");
    FFB6D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1k) | reinterpret_cast<uintptr_t>(w2k) |
                    reinterpret_cast<uintptr_t>(w3k)) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & (4 * SZ - 1)) == 0,
                  "mlp_chain3_pm: x and the weights must be 16-byte aligned, out to four channels");
    ChainParams p;
    p.x = x; p.w1 = w1k; p.w2 = w2k; p.w3 = w3k; p.b1 = b1; p.b2 = b2; p.b3 = b3;
    p.s1 = slope_of(act1); p.s2 = slope_of(act2); p.s3 = slope_of(act3);
    p.out = out; p.ldx = (int)ldx; p.ldo = (int)ldo; p.cout3 = (int)cout3; p.rows = (int)rows;
    p.n_tiles = (int)ceil_div(rows, 128);
    static int attr_set[kMaxDevices + 1];                                      // per device (common.h: device_slot)
    const int slot = device_slot();
    if (!cache_get(attr_set, slot)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain3_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Chain<T>::LDS) != hipSuccess)
            return set_error(FFB6D_ERR_HIP, "mlp_chain3_pm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        cache_set(attr_set, slot, 1);
    }
    static int cu_count[kMaxDevices + 1];                                   // workgroups per CU by LDS: one in fp32 (145 KB), two in bf16 (73 KB)
    int cus = cache_get(cu_count, slot);
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        cache_set(cu_count, slot, cus);
    }
    const unsigned grid = (unsigned)std::min<int64_t>(p.n_tiles, (int64_t)cus * (SZ == 2 ? 2 : 1));
    hipLaunchKernelGGL((mlp_chain3_kernel<T>), dim3(grid), dim3(BLK), Chain<T>::LDS, as_stream(stream), p);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // namespace
}  // namespace ffb6d

extern "C" int ffb6d_mlp_chain3_pm_f32(const float* x, int64_t ldx, const float* w1k, const float* b1, int act1, const float* w2k,
                                       const float* b2, int act2, const float* w3k, const float* b3, int act3, float* out, int64_t ldo,
                                       int64_t rows, int64_t cout3, ffb6d_stream_t stream)
{
    return mlp_chain3_impl<float>(x, ldx, w1k, b1, act1, w2k, b2, act2, w3k, b3, act3, out, ldo, rows, cout3, stream);
}

extern "C" int ffb6d_mlp_chain3_pm_bf16(const void* x, int64_t ldx, const void* w1k, const float* b1, int act1, const void* w2k,
                                        const float* b2, int act2, const void* w3k, const float* b3, int act3, void* out, int64_t ldo,
                                        int64_t rows, int64_t cout3, ffb6d_stream_t stream)
{
    return mlp_chain3_impl<__bf16>(x, ldx, w1k, b1, act1, w2k, b2, act2, w3k, b3, act3, out, ldo, rows, cout3, stream);
}
