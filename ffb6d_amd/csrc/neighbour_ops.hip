// ffb6d_amd/csrc/neighbour_ops.hip -- RandLA-Net neighbour ops + pixel<->point fusion
// gathers for gfx950, in the reference's own tensor layouts (drop-in operators).
//
// Reference bodies (all stock torch ops there; see include/ffb6d_ops.h for the map):
//   FFB6D.random_sample          ffb6d/models/ffb6d.py:159-177   gather K + max over K
//   FFB6D.nearest_interpolation  ffb6d/models/ffb6d.py:179-194   1-NN gather
//   gather_neighbour             ffb6d/models/RandLA/RandLANet.py:225-234
//   relative_pos_encoding        ffb6d/models/RandLA/RandLANet.py:216-223
//   Att_pooling (softmax*feat, sum over K)    RandLANet.py:245-248
//
// All of these are HBM-bound gathers: the kernels below read every index once, keep
// it in registers across the channel loop (the reference materialises an int64
// [B,C,N*K] index tensor instead, ffb6d.py:171-173), put the lane dimension on the
// contiguous axis of the OUTPUT so every store is a full coalesced wave store, and use
// 16-byte accesses wherever the layout allows.
#include "common.h"
#include "ffb6d_ops.h"

#include <cfloat>

namespace ffb6d {
bool nearest_interp_bwd_lds(const float* grad_out, const void* idx, int idx_bits, float* grad_feat, int64_t B, int64_t C, int64_t M,
                            int64_t U, hipStream_t st);      // csrc/train_ops.hip

namespace {

constexpr int BLK = 256;

// ------------------------------------------------------------------------------------
// random_sample: out[b,c,n] = max_k feat[b,c,idx[b,n,k]]
// lanes run over n (contiguous in out and in idx rows); each lane keeps its K indices in
// registers and walks `cc` channels; grid = (ceil(Np/256), ceil(C/cc), B).
// ------------------------------------------------------------------------------------
template <typename IdxT, int K>
__global__ void __launch_bounds__(BLK)
random_sample_kernel(const float* __restrict__ feat, const IdxT* __restrict__ idx,
                     float* __restrict__ out, int32_t* __restrict__ arg, int C, int M, int Np,
                     int cc)
{
    const int n = blockIdx.x * BLK + threadIdx.x;
    const int b = blockIdx.z;
    if (n >= Np) return;
    int ii[K];
    const IdxT* ip = idx + ((size_t)b * Np + n) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) ii[k] = (int)ip[k];
    const int c0 = blockIdx.y * cc;
    const int c1 = min(C, c0 + cc);
    for (int c = c0; c < c1; ++c) {
        const float* row = feat + ((size_t)b * C + c) * M;
        float v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = row[ii[k]];
        float m = v[0];
        int am = ii[0];
#pragma unroll
        for (int k = 1; k < K; ++k)
            if (v[k] > m || v[k] != v[k]) { m = v[k]; am = ii[k]; }     // torch.max: NaN propagates
        const size_t o = ((size_t)b * C + c) * Np + n;
        out[o] = m;
        if (arg) arg[o] = am;
    }
}

// K = 16, inference (no arg-max needed): a ROW of 16 lanes owns one output point, lane r gathers
// neighbour r.  For the pixel->point fusion the 16 neighbours of a point are a ~4x4 pixel patch, so
// the 16 addresses of a row fall into 4-5 cache lines instead of 16 different ones (one lane per
// point makes every lane of a gather hit its own line); the max over the row is four row_ror DPP
// steps.  Lane (c & 15) keeps the result of channel c, so results leave as one store per 16 channels.
__device__ __forceinline__ float nan_max(float a, float b)   // torch.max semantics: a NaN operand wins
{
    return (a != a) ? a : ((b > a || b != b) ? b : a);
}

__device__ __forceinline__ float row16_max_dpp(float v)
{
    v = nan_max(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));  // row_ror:8
    v = nan_max(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));  // row_ror:4
    v = nan_max(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));  // row_ror:2
    v = nan_max(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));  // row_ror:1
    return v;
}

template <typename IdxT>
__global__ void __launch_bounds__(BLK)
random_sample_row16_kernel(const float* __restrict__ feat, const IdxT* __restrict__ idx,
                           float* __restrict__ out, int C, int M, int Np, int cc)
{
    const int r = threadIdx.x & 15;
    const int n = blockIdx.x * (BLK / 16) + (threadIdx.x >> 4);
    const int b = blockIdx.z;
    const bool valid = n < Np;
    const int my = valid ? (int)idx[((size_t)b * Np + n) * 16 + r] : 0;
    const int c0 = blockIdx.y * cc;
    const int c1 = min(C, c0 + cc);
    const float* base = feat + (size_t)b * C * M + my;
    float* obase = out + (size_t)b * C * Np + n;
    for (int cb = c0; cb < c1; cb += 16) {
        float keep = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = cb + j;
            float v = -INFINITY;
            if (c < c1) v = base[(size_t)c * M];
            const float m = row16_max_dpp(v);
            if (j == r) keep = m;
        }
        const int c = cb + r;
        if (valid && c < c1) obase[(size_t)c * Np] = keep;
    }
}

// any K (indices re-read per channel; they stay L1/L2 resident)
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
random_sample_anyk_kernel(const float* __restrict__ feat, const IdxT* __restrict__ idx,
                          float* __restrict__ out, int32_t* __restrict__ arg, int C, int M,
                          int Np, int K, int cc)
{
    const int n = blockIdx.x * BLK + threadIdx.x;
    const int b = blockIdx.z;
    if (n >= Np) return;
    const IdxT* ip = idx + ((size_t)b * Np + n) * K;
    const int c0 = blockIdx.y * cc;
    const int c1 = min(C, c0 + cc);
    for (int c = c0; c < c1; ++c) {
        const float* row = feat + ((size_t)b * C + c) * M;
        int am = (int)ip[0];
        float m = row[am];
        for (int k = 1; k < K; ++k) {
            const int i = (int)ip[k];
            const float v = row[i];
            if (v > m || v != v) { m = v; am = i; }                    // torch.max: NaN propagates
        }
        const size_t o = ((size_t)b * C + c) * Np + n;
        out[o] = m;
        if (arg) arg[o] = am;
    }
}

__global__ void __launch_bounds__(BLK)
random_sample_bwd_kernel(const float* __restrict__ grad_out, const int32_t* __restrict__ arg,
                         float* __restrict__ grad_feat, int M, int Np, size_t rows)
{
    // rows = B*C ; one thread per (row, n)
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= rows * (size_t)Np) return;
    const size_t row = t / Np;
    unsafeAtomicAdd(grad_feat + row * M + arg[t], grad_out[t]);
}

// ------------------------------------------------------------------------------------
// nearest_interpolation: out[b,c,u] = feat[b,c,idx[b,u]]
// V consecutive u per lane (V=4: one 16-byte store per channel), cc channels per block.
// ------------------------------------------------------------------------------------
template <typename IdxT, int V>
__global__ void __launch_bounds__(BLK)
nearest_interp_kernel(const float* __restrict__ feat, const IdxT* __restrict__ idx,
                      float* __restrict__ out, int C, int M, int U, int cc)
{
    const int u0 = (blockIdx.x * BLK + threadIdx.x) * V;
    const int b = blockIdx.z;
    if (u0 >= U) return;
    int ii[V];
#pragma unroll
    for (int v = 0; v < V; ++v) ii[v] = (int)idx[(size_t)b * U + min(u0 + v, U - 1)];
    const int c0 = blockIdx.y * cc;
    const int c1 = min(C, c0 + cc);
    for (int c = c0; c < c1; ++c) {
        const float* row = feat + ((size_t)b * C + c) * M;
        float* orow = out + ((size_t)b * C + c) * U + u0;
        if constexpr (V == 4) {
            float4 r = make_float4(row[ii[0]], row[ii[1]], row[ii[2]], row[ii[3]]);
            *reinterpret_cast<float4*>(orow) = r;  // U % 4 == 0 guaranteed by the launcher
        } else {
            orow[0] = row[ii[0]];
        }
    }
}

template <typename IdxT>
__global__ void __launch_bounds__(BLK)
nearest_interp_bwd_kernel(const float* __restrict__ grad_out, const IdxT* __restrict__ idx,
                          float* __restrict__ grad_feat, int C, int M, int U, size_t total)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;  // over B*C*U
    if (t >= total) return;
    const size_t row = t / U;       // b*C + c
    const int u = (int)(t - row * U);
    const size_t b = row / C;
    const int i = (int)idx[b * U + u];
    unsafeAtomicAdd(grad_feat + row * M + i, grad_out[t]);
}

// ------------------------------------------------------------------------------------
// gather_neighbour: out[b,n,k,:] = pc[b,idx[b,n,k],:]   (point-major rows of C floats)
// ------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
gather_rows_vec4_kernel(const float4* __restrict__ pc, const IdxT* __restrict__ idx,
                        float4* __restrict__ out, int M, int C4, size_t rows_per_b, size_t total)
{
    // one lane per float4 of an output row; total = B*N*K*C4
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / C4;
    const int c = (int)(t - row * C4);
    const size_t b = row / rows_per_b;
    const int src = (int)idx[row];
    out[t] = pc[(b * M + src) * C4 + c];
}

template <typename IdxT>
__global__ void __launch_bounds__(BLK)
gather_rows_scalar_kernel(const float* __restrict__ pc, const IdxT* __restrict__ idx,
                          float* __restrict__ out, int M, int C, size_t rows_per_b, size_t total)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / C;
    const int c = (int)(t - row * C);
    const size_t b = row / rows_per_b;
    const int src = (int)idx[row];
    out[t] = pc[(b * M + src) * C + c];
}

template <typename IdxT>
__global__ void __launch_bounds__(BLK)
gather_rows_bwd_kernel(const float* __restrict__ grad_out, const IdxT* __restrict__ idx,
                       float* __restrict__ grad_pc, int M, int C, size_t rows_per_b, size_t total)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / C;
    const int c = (int)(t - row * C);
    const size_t b = row / rows_per_b;
    const int src = (int)idx[row];
    unsafeAtomicAdd(grad_pc + (b * M + src) * C + c, grad_out[t]);
}

// ------------------------------------------------------------------------------------
// relative_pos_encoding: out[b,n,k,:] = [dis, p-q, p, q]  (10 floats per (n,k) pair)
// one lane per pair; the 256x10 block result is transposed through LDS so the global
// store is 640 contiguous float4.
// ------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
rel_pos_enc_kernel(const float* __restrict__ xyz, const IdxT* __restrict__ idx,
                   float* __restrict__ out, int N, int K, size_t total)
{
    __shared__ __attribute__((aligned(16))) float stage[BLK * 10];
    const size_t t0 = (size_t)blockIdx.x * BLK;
    const size_t t = t0 + threadIdx.x;  // pair id over B*N*K
    if (t < total) {
        const size_t pn = t / K;         // b*N + n
        const size_t b = pn / N;
        const int j = (int)idx[t];
        const float* p = xyz + pn * 3;
        const float* q = xyz + (b * N + j) * 3;
        const float px = p[0], py = p[1], pz = p[2];
        const float qx = q[0], qy = q[1], qz = q[2];
        const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        float* o = stage + threadIdx.x * 10;
        o[0] = __fsqrt_rn(s);
        o[1] = dx; o[2] = dy; o[3] = dz;
        o[4] = px; o[5] = py; o[6] = pz;
        o[7] = qx; o[8] = qy; o[9] = qz;
    }
    __syncthreads();
    const size_t remain = total - t0;  // pairs in this block (>= 1)
    const int npairs = remain < (size_t)BLK ? (int)remain : BLK;
    float* gout = out + t0 * 10;       // 10240-byte aligned per block
    if (npairs == BLK) {
        const float4* s4 = reinterpret_cast<const float4*>(stage);
        float4* g4 = reinterpret_cast<float4*>(gout);
        for (int i = threadIdx.x; i < BLK * 10 / 4; i += BLK) g4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < npairs * 10; i += BLK) gout[i] = stage[i];
    }
}

// channel-major variant: out[b, ch, n, k], ch = 0..9 -- the layout the first LFA MLP consumes
// (the reference permutes the point-major tensor, RandLANet.py:198); lanes run over (n,k) pairs,
// so every one of the 10 channel stores is a full coalesced wave store.
template <typename IdxT>
__global__ void __launch_bounds__(BLK)
rel_pos_enc_cm_kernel(const float* __restrict__ xyz, const IdxT* __restrict__ idx,
                      float* __restrict__ out, int N, int K, size_t per_b /* N*K */, size_t total)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;  // pair id over B*N*K
    if (t >= total) return;
    const size_t b = t / per_b;
    const size_t nk = t - b * per_b;
    const size_t n = nk / K;
    const int j = (int)idx[t];
    const float* p = xyz + (b * N + n) * 3;
    const float* q = xyz + (b * N + j) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float qx = q[0], qy = q[1], qz = q[2];
    const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    float* o = out + b * 10 * per_b + nk;
    o[0 * per_b] = __fsqrt_rn(s);
    o[1 * per_b] = dx; o[2 * per_b] = dy; o[3 * per_b] = dz;
    o[4 * per_b] = px; o[5 * per_b] = py; o[6 * per_b] = pz;
    o[7 * per_b] = qx; o[8 * per_b] = qy; o[9 * per_b] = qz;
}

// ------------------------------------------------------------------------------------
// att_pool: rows of K contiguous floats; LPR lanes per row, one float4 per lane
// (K = 4*LPR).  softmax max/sum and the weighted sum are wave-shuffle reductions inside
// the LPR-lane group.
// ------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LPR>
__global__ void __launch_bounds__(BLK)
att_pool_kernel(const float4* __restrict__ feat, const float4* __restrict__ act,
                float* __restrict__ out, size_t rows)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;  // float4 id = row*LPR + lane
    const size_t row = t / LPR;
    const bool live = row < rows;
    float4 f = make_float4(0, 0, 0, 0), a = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    if (live) { f = feat[t]; a = act[t]; }
    const float m = group_max<LPR>(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    const float e0 = expf(a.x - m), e1 = expf(a.y - m), e2 = expf(a.z - m), e3 = expf(a.w - m);
    const float den = group_sum<LPR>((e0 + e1) + (e2 + e3));
    // torch: softmax = exp(x-max)/sum, then feat*score, then sum over K (RandLANet.py:246-248)
    float acc = f.x * (e0 / den);
    acc += f.y * (e1 / den);
    acc += f.z * (e2 / den);
    acc += f.w * (e3 / den);
    acc = group_sum<LPR>(acc);
    if (live && (threadIdx.x % LPR) == 0) out[row] = acc;
}


// generic K: one lane per row
__global__ void __launch_bounds__(BLK)
att_pool_anyk_kernel(const float* __restrict__ feat, const float* __restrict__ act,
                     float* __restrict__ out, size_t rows, int K)
{
    const size_t row = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (row >= rows) return;
    const float* f = feat + row * K;
    const float* a = act + row * K;
    float m = a[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, a[k]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(a[k] - m);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += f[k] * (expf(a[k] - m) / den);
    out[row] = acc;
}

// backward: gf = g*s ; ga = g*s*(f - y)
__global__ void __launch_bounds__(BLK)
att_pool_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ feat,
                    const float* __restrict__ act, float* __restrict__ grad_feat,
                    float* __restrict__ grad_act, size_t rows, int K)
{
    const size_t row = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (row >= rows) return;
    const float* f = feat + row * K;
    const float* a = act + row * K;
    float m = a[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, a[k]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(a[k] - m);
    float y = 0.f;
    for (int k = 0; k < K; ++k) y += f[k] * (expf(a[k] - m) / den);
    const float g = grad_out[row];
    for (int k = 0; k < K; ++k) {
        const float s = expf(a[k] - m) / den;
        grad_feat[row * K + k] = g * s;
        grad_act[row * K + k] = g * s * (f[k] - y);
    }
}

// K = 16 (the network's neighbourhood size): a row is one 64-byte line -- four 16-byte loads per operand, every exponential
// computed once, four 16-byte stores per gradient: the pass moves its 4 x 64 bytes per row at the HBM rate (the generic kernel
// above reads a row as 16 scalar loads per lane, 64 lines per instruction, and evaluates every exponential three times)
__global__ void __launch_bounds__(BLK)
att_pool_bwd16_kernel(const float* __restrict__ grad_out, const float4* __restrict__ feat, const float4* __restrict__ act,
                      float4* __restrict__ grad_feat, float4* __restrict__ grad_act, size_t rows)
{
    const size_t row = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (row >= rows) return;
    float f[16], e[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 fv = feat[row * 4 + q], av = act[row * 4 + q];
        f[4 * q] = fv.x; f[4 * q + 1] = fv.y; f[4 * q + 2] = fv.z; f[4 * q + 3] = fv.w;
        e[4 * q] = av.x; e[4 * q + 1] = av.y; e[4 * q + 2] = av.z; e[4 * q + 3] = av.w;
    }
    float m = e[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, e[k]);
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { e[k] = expf(e[k] - m); den += e[k]; }
    float y = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { e[k] = e[k] / den; y += f[k] * e[k]; }       // the generic kernel's arithmetic: s = exp / den
    const float g = grad_out[row];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float gf[4], ga[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gs = g * e[4 * q + j];
            gf[j] = gs;
            ga[j] = gs * (f[4 * q + j] - y);
        }
        grad_feat[row * 4 + q] = make_float4(gf[0], gf[1], gf[2], gf[3]);
        grad_act[row * 4 + q] = make_float4(ga[0], ga[1], ga[2], ga[3]);
    }
}

template <typename IdxT>
__global__ void __launch_bounds__(BLK)
check_range_kernel(const IdxT* __restrict__ idx, size_t count, long long M, int32_t* bad)
{
    const size_t t = (size_t)blockIdx.x * BLK + threadIdx.x;
    int mine = 0;
    if (t < count) {
        const long long v = (long long)idx[t];
        mine = (v < 0 || v >= M) ? 1 : 0;
    }
    const unsigned long long ball = __ballot(mine);
    if ((threadIdx.x & 63) == 0 && ball) atomicAdd(bad, (int)__popcll(ball));
}

// channels per block so that the grid has roughly >= 2048 blocks
int pick_cc(int64_t tiles, int64_t B, int64_t C)
{
    const int64_t target = 2048;
    int64_t chunks = ceil_div(target, tiles * B);
    if (chunks < 1) chunks = 1;
    if (chunks > C) chunks = C;
    int64_t cc = ceil_div(C, chunks);
    return (int)(cc < 1 ? 1 : cc);
}

bool bits_ok(int bits) { return bits == 32 || bits == 64; }

}  // namespace
}  // namespace ffb6d

using namespace ffb6d;

#define DISPATCH_IDX(bits, IdxT, ...)                       \
    if ((bits) == 64) { using IdxT = int64_t; __VA_ARGS__ } \
    else { using IdxT = int32_t; __VA_ARGS__ }

extern "C" {

int ffb6d_random_sample_f32(const float* feat, const void* idx, int idx_bits, float* out,
                            int32_t* arg, int64_t B, int64_t C, int64_t M, int64_t Np, int K,
                            ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "random_sample: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 0 && Np >= 0 && M >= 1 && K >= 1, "random_sample: bad shape");
    FFB6D_REQUIRE(M < (1LL << 31) && Np < (1LL << 31) && B < 65536 && C < (1LL << 31),
                  "random_sample: size too large");
    if (B == 0 || C == 0 || Np == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && out, "random_sample: null pointer");
    const int64_t tiles = ceil_div(Np, BLK);
    const int cc = pick_cc(tiles, B, C);
    dim3 grid((unsigned)tiles, (unsigned)ceil_div(C, cc), (unsigned)B);
    hipStream_t st = as_stream(stream);
    DISPATCH_IDX(idx_bits, IdxT, {
        const IdxT* ip = static_cast<const IdxT*>(idx);
        // the row layout pays when the 16 neighbours of a point are close in memory (image sources:
        // M = h*w >> Np); for the shuffled cloud (pooling a level onto its first quarter, M = 4*Np) every
        // neighbour sits in its own line either way and one lane per point issues fewer reductions
        if (K == 16 && arg == nullptr && M > 4 * Np) {
            const int64_t rtiles = ceil_div(Np, BLK / 16);
            int rcc = pick_cc(rtiles, B, C);
            rcc = (rcc + 15) / 16 * 16;                        // whole groups of 16 channels
            dim3 rgrid((unsigned)rtiles, (unsigned)ceil_div(C, rcc), (unsigned)B);
            hipLaunchKernelGGL((random_sample_row16_kernel<IdxT>), rgrid, dim3(BLK), 0, st, feat, ip, out,
                               (int)C, (int)M, (int)Np, rcc);
        } else if (K == 16)
            hipLaunchKernelGGL((random_sample_kernel<IdxT, 16>), grid, dim3(BLK), 0, st, feat, ip,
                               out, arg, (int)C, (int)M, (int)Np, cc);
        else
            hipLaunchKernelGGL((random_sample_anyk_kernel<IdxT>), grid, dim3(BLK), 0, st, feat, ip,
                               out, arg, (int)C, (int)M, (int)Np, K, cc);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_random_sample_bwd_f32(const float* grad_out, const int32_t* arg, float* grad_feat,
                                int64_t B, int64_t C, int64_t M, int64_t Np,
                                ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && C >= 0 && Np >= 0 && M >= 1, "random_sample_bwd: bad shape");
    if (B == 0 || C == 0) return FFB6D_OK;
    FFB6D_REQUIRE(grad_out && arg && grad_feat, "random_sample_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    FFB6D_HIP_TRY(hipMemsetAsync(grad_feat, 0, (size_t)B * C * M * sizeof(float), st));
    const size_t total = (size_t)B * C * Np;
    if (total == 0) return FFB6D_OK;
    hipLaunchKernelGGL(random_sample_bwd_kernel, dim3((unsigned)ceil_div(total, BLK)), dim3(BLK), 0,
                       st, grad_out, arg, grad_feat, (int)M, (int)Np, (size_t)B * C);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_nearest_interpolation_f32(const float* feat, const void* idx, int idx_bits, float* out,
                                    int64_t B, int64_t C, int64_t M, int64_t U,
                                    ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "nearest_interpolation: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 0 && U >= 0 && M >= 1, "nearest_interpolation: bad shape");
    FFB6D_REQUIRE(M < (1LL << 31) && U < (1LL << 31) && B < 65536, "nearest_interpolation: too large");
    if (B == 0 || C == 0 || U == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && idx && out, "nearest_interpolation: null pointer");
    hipStream_t st = as_stream(stream);
    const bool vec = (U % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const int V = vec ? 4 : 1;
    const int64_t tiles = ceil_div(U, (int64_t)BLK * V);
    const int cc = pick_cc(tiles, B, C);
    dim3 grid((unsigned)tiles, (unsigned)ceil_div(C, cc), (unsigned)B);
    DISPATCH_IDX(idx_bits, IdxT, {
        const IdxT* ip = static_cast<const IdxT*>(idx);
        if (vec)
            hipLaunchKernelGGL((nearest_interp_kernel<IdxT, 4>), grid, dim3(BLK), 0, st, feat, ip,
                               out, (int)C, (int)M, (int)U, cc);
        else
            hipLaunchKernelGGL((nearest_interp_kernel<IdxT, 1>), grid, dim3(BLK), 0, st, feat, ip,
                               out, (int)C, (int)M, (int)U, cc);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_nearest_interpolation_bwd_f32(const float* grad_out, const void* idx, int idx_bits,
                                        float* grad_feat, int64_t B, int64_t C, int64_t M,
                                        int64_t U, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "nearest_interpolation_bwd: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 0 && U >= 0 && M >= 1, "nearest_interpolation_bwd: bad shape");
    if (B == 0 || C == 0) return FFB6D_OK;
    FFB6D_REQUIRE(grad_out && idx && grad_feat, "nearest_interpolation_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    // rows of one frame and a group of channels privatised in LDS (csrc/train_ops.hip) whenever M rows of floats fit: no global
    // atomics, no memset; else the global scatter-add below
    if (U > 0 && nearest_interp_bwd_lds(grad_out, idx, idx_bits, grad_feat, B, C, M, U, st)) {
        FFB6D_LAUNCH_CHECK();
        return FFB6D_OK;
    }
    FFB6D_HIP_TRY(hipMemsetAsync(grad_feat, 0, (size_t)B * C * M * sizeof(float), st));
    const size_t total = (size_t)B * C * U;
    if (total == 0) return FFB6D_OK;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((nearest_interp_bwd_kernel<IdxT>), dim3((unsigned)ceil_div(total, BLK)),
                           dim3(BLK), 0, st, grad_out, static_cast<const IdxT*>(idx), grad_feat,
                           (int)C, (int)M, (int)U, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_gather_neighbour_f32(const float* pc, const void* idx, int idx_bits, float* out,
                               int64_t B, int64_t M, int64_t C, int64_t N, int K,
                               ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "gather_neighbour: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 1 && K >= 1, "gather_neighbour: bad shape");
    if (B == 0 || C == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(pc && idx && out, "gather_neighbour: null pointer");
    hipStream_t st = as_stream(stream);
    const size_t rows_per_b = (size_t)N * K;
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(pc) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    DISPATCH_IDX(idx_bits, IdxT, {
        const IdxT* ip = static_cast<const IdxT*>(idx);
        if (vec) {
            const int C4 = (int)(C / 4);
            const size_t total = (size_t)B * rows_per_b * C4;
            hipLaunchKernelGGL((gather_rows_vec4_kernel<IdxT>), dim3((unsigned)ceil_div(total, BLK)),
                               dim3(BLK), 0, st, reinterpret_cast<const float4*>(pc), ip,
                               reinterpret_cast<float4*>(out), (int)M, C4, rows_per_b, total);
        } else {
            const size_t total = (size_t)B * rows_per_b * C;
            hipLaunchKernelGGL((gather_rows_scalar_kernel<IdxT>), dim3((unsigned)ceil_div(total, BLK)),
                               dim3(BLK), 0, st, pc, ip, out, (int)M, (int)C, rows_per_b, total);
        }
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_gather_neighbour_bwd_f32(const float* grad_out, const void* idx, int idx_bits,
                                   float* grad_pc, int64_t B, int64_t M, int64_t C, int64_t N,
                                   int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "gather_neighbour_bwd: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 1 && K >= 1, "gather_neighbour_bwd: bad shape");
    if (B == 0 || C == 0) return FFB6D_OK;
    FFB6D_REQUIRE(grad_out && idx && grad_pc, "gather_neighbour_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    FFB6D_HIP_TRY(hipMemsetAsync(grad_pc, 0, (size_t)B * M * C * sizeof(float), st));
    const size_t rows_per_b = (size_t)N * K;
    const size_t total = (size_t)B * rows_per_b * C;
    if (total == 0) return FFB6D_OK;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((gather_rows_bwd_kernel<IdxT>), dim3((unsigned)ceil_div(total, BLK)),
                           dim3(BLK), 0, st, grad_out, static_cast<const IdxT*>(idx), grad_pc, (int)M,
                           (int)C, rows_per_b, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_relative_pos_encoding_f32(const float* xyz, const void* idx, int idx_bits, float* out,
                                    int64_t B, int64_t N, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "relative_pos_encoding: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && N >= 0 && K >= 1, "relative_pos_encoding: bad shape");
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(xyz && idx && out, "relative_pos_encoding: null pointer");
    FFB6D_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "relative_pos_encoding: out must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const size_t total = (size_t)B * N * K;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((rel_pos_enc_kernel<IdxT>), dim3((unsigned)ceil_div(total, BLK)), dim3(BLK),
                           0, st, xyz, static_cast<const IdxT*>(idx), out, (int)N, K, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_att_pool_f32(const float* feat, const float* act, float* out, int64_t B, int64_t C,
                       int64_t N, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && K >= 1, "att_pool: bad shape");
    const size_t rows = (size_t)B * C * N;
    if (rows == 0) return FFB6D_OK;
    FFB6D_REQUIRE(feat && act && out, "att_pool: null pointer");
    hipStream_t st = as_stream(stream);
    const bool aligned = ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(act)) & 15) == 0;
    const float4* f4 = reinterpret_cast<const float4*>(feat);
    const float4* a4 = reinterpret_cast<const float4*>(act);
    if (aligned && K == 16) {
        hipLaunchKernelGGL((att_pool_kernel<4>), dim3((unsigned)ceil_div(rows * 4, BLK)), dim3(BLK), 0,
                           st, f4, a4, out, rows);
    } else if (aligned && K == 32) {
        hipLaunchKernelGGL((att_pool_kernel<8>), dim3((unsigned)ceil_div(rows * 8, BLK)), dim3(BLK), 0,
                           st, f4, a4, out, rows);
    } else if (aligned && K == 8) {
        hipLaunchKernelGGL((att_pool_kernel<2>), dim3((unsigned)ceil_div(rows * 2, BLK)), dim3(BLK), 0,
                           st, f4, a4, out, rows);
    } else if (aligned && K == 4) {
        hipLaunchKernelGGL((att_pool_kernel<1>), dim3((unsigned)ceil_div(rows, BLK)), dim3(BLK), 0, st,
                           f4, a4, out, rows);
    } else {
        hipLaunchKernelGGL(att_pool_anyk_kernel, dim3((unsigned)ceil_div(rows, BLK)), dim3(BLK), 0, st,
                           feat, act, out, rows, K);
    }
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_relative_pos_encoding_cm_f32(const float* xyz, const void* idx, int idx_bits, float* out,
                                       int64_t B, int64_t N, int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "relative_pos_encoding_cm: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(B >= 0 && N >= 0 && K >= 1, "relative_pos_encoding_cm: bad shape");
    if (B == 0 || N == 0) return FFB6D_OK;
    FFB6D_REQUIRE(xyz && idx && out, "relative_pos_encoding_cm: null pointer");
    hipStream_t st = as_stream(stream);
    const size_t per_b = (size_t)N * K, total = (size_t)B * per_b;
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((rel_pos_enc_cm_kernel<IdxT>), dim3((unsigned)ceil_div((int64_t)total, BLK)), dim3(BLK), 0,
                           st, xyz, static_cast<const IdxT*>(idx), out, (int)N, K, per_b, total);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_att_pool_bwd_f32(const float* grad_out, const float* feat, const float* act,
                           float* grad_feat, float* grad_act, int64_t B, int64_t C, int64_t N,
                           int K, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(B >= 0 && C >= 0 && N >= 0 && K >= 1, "att_pool_bwd: bad shape");
    const size_t rows = (size_t)B * C * N;
    if (rows == 0) return FFB6D_OK;
    FFB6D_REQUIRE(grad_out && feat && act && grad_feat && grad_act, "att_pool_bwd: null pointer");
    const bool aligned = ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(grad_feat) |
                           reinterpret_cast<uintptr_t>(grad_act)) & 15) == 0;
    if (K == 16 && aligned)
        hipLaunchKernelGGL(att_pool_bwd16_kernel, dim3((unsigned)ceil_div(rows, BLK)), dim3(BLK), 0, as_stream(stream), grad_out,
                           reinterpret_cast<const float4*>(feat), reinterpret_cast<const float4*>(act),
                           reinterpret_cast<float4*>(grad_feat), reinterpret_cast<float4*>(grad_act), rows);
    else
        hipLaunchKernelGGL(att_pool_bwd_kernel, dim3((unsigned)ceil_div(rows, BLK)), dim3(BLK), 0,
                           as_stream(stream), grad_out, feat, act, grad_feat, grad_act, rows, K);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

int ffb6d_check_index_range(const void* idx, int idx_bits, int64_t count, int64_t M, int32_t* bad,
                            ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(bits_ok(idx_bits), "check_index_range: idx_bits must be 32 or 64");
    FFB6D_REQUIRE(bad, "check_index_range: null counter");
    hipStream_t st = as_stream(stream);
    FFB6D_HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
    if (count <= 0) return FFB6D_OK;
    FFB6D_REQUIRE(idx, "check_index_range: null pointer");
    DISPATCH_IDX(idx_bits, IdxT, {
        hipLaunchKernelGGL((check_range_kernel<IdxT>), dim3((unsigned)ceil_div(count, BLK)), dim3(BLK),
                           0, st, static_cast<const IdxT*>(idx), (size_t)count, (long long)M, bad);
    })
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // extern "C"
