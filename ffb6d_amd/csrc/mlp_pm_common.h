// ffb6d_amd/csrc/mlp_pm_common.h -- launch parameters and the shared epilogue of the point-major shared-MLP GEMM kernels
// (csrc/mlp_pm.hip, csrc/mlp_pm_big.hip).  See csrc/mlp_pm.hip for the layout and the reference citations.
#pragma once
#include "common.h"
#include "mfma_pm.h"

namespace ffb6d {
namespace pm {

// bytes a row buffer [rows][ld] spans when only the first k elements of its last row belong to it (the operand may be a channel
// slice of a wider row buffer: the descriptor must not reach past the slice's last element)
__device__ __forceinline__ unsigned span_bytes(unsigned rows, int ld, int k, int sz)
{
    return rows ? ((rows - 1u) * (unsigned)ld + (unsigned)k) * (unsigned)sz : 0u;
}

struct PmParams {
    const void* w;        // [cout, k1 + k2]  (nn.Conv weight layout, BatchNorm folded), element type T
    const float* bias;    // [cout] fp32 or null
    const void* x1;       // [rows, ld1] (or [B * px, ld1] when xidx), first k1 elements of a row are used
    const void* xidx;     // [rows] int32/int64 or null: x1 row of output row r = (r / P) * px + xidx[r]  (operand gather)
    const void* x2;       // [rows, ld2] or null
    const void* y;        // [B * py, ldy] rows added in the epilogue, or null
    const void* gidx;     // [rows] int32/int64: row of the frame's py rows to add; null with y != null: row r itself
    void* out;            // [rows, ldo]
    int rows, cout, k1, k2, ld1, ld2, ldy, ldo;
    int P, py, px;        // output rows per frame (for the gathers), rows of Y / of X1 per frame
    int act, idx64;
    int n_pt, n_ct;       // point tiles, channel tiles
    // tile-sequence form: three regions of point tiles with decreasing sequence lengths (guided schedule: the last workgroups the
    // dispatcher hands out are short ones).  Region A = point tiles [0, pt_b): tpg tiles per workgroup; B = [pt_b, pt_c): tpg_b;
    // C = [pt_c, n_pt): one tile per workgroup.  wg_b / wg_c = first workgroup of regions B / C.
    int tpg, tpg_b, pt_b, pt_c, wg_b, wg_c;
    // ... or (round 6) lin_wg > 0: every XCD's tile list cut into lin_wg balanced contiguous sequences (mlp_pm_seq_kernel: LIN)
    int lin_wg = 0;
};

// epilogue shared by the GEMM kernels: bias, gathered / added row of Y, activation or log-softmax, store
// LSM = false compiles the log-softmax branch out (it needs all channels of a point live at once: 16 * TM more registers)
// ypre (use_pre): the gathered / added rows of Y fetched ahead by the caller, [j][i][g]; an array REFERENCE with static indices and a flag
// -- a pointer that may be null would put the array into scratch memory
template <typename T, int TM, int TN, bool LSM = true>
__device__ __forceinline__ void pm_epilogue(const PmParams& p, f32x16 (&acc)[TM][TN], int c0, int r0, int wm, int wn, int l31, int kh,
                                            const typename El<T>::Raw4 (&ypre)[TM * TN * 4], bool use_pre)
{
    constexpr int SZ = El<T>::SZ;
    // epilogue: lane = one point, 4 groups of 4 consecutive channels per 32 x 32 tile
    const T* yb = static_cast<const T*>(p.y);
    T* ob = static_cast<T*>(p.out);
    const float slope = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : 0.2f);
    constexpr int AL = 4 * SZ - 1;                         // alignment mask of a 4-channel group
    const bool vec = (p.cout & 3) == 0 && (p.ldo & 3) == 0 && (p.ldy & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.y)) & AL) == 0 &&
                     (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = r0 + (wn * TN + j) * 32 + l31;
        const bool live = r < p.rows;
        const T* yrow = nullptr;
        if (yb && live) {
            long long yr = r;
            if (p.gidx) {
                const long long gi = p.idx64 ? static_cast<const long long*>(p.gidx)[r]
                                             : (long long)static_cast<const int*>(p.gidx)[r];
                yr = (long long)(r / p.P) * p.py + gi;
            }
            yrow = yb + yr * p.ldy;
        }
        T* orow = ob + (size_t)r * p.ldo;
        if constexpr (SZ == 2) {
            // bf16: a point's 8 consecutive channels sit in lanes l (channels 8g .. 8g+3) and l + 32 (8g+4 .. 8g+7): four 8-byte
            // stores per 32 x 32 tile and lane.  The epilogue's time is per store INSTRUCTION (~100 cycles per CU each, whatever the
            // width: profiles/r04_gemm_epilogue_probe_bf16.txt -- half of the bf16 GEMMs' time), so pairs of channel groups trade
            // halves across the half-waves (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second) and
            // every lane stores 16 bytes: the lower lane all 8 channels of group g, the upper lane all 8 of group g + 1.
            const bool wide = vec && !(LSM && p.act == 3) && (p.cout & 15) == 0 && (p.ldo & 7) == 0 &&
                              (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
            if (wide) {                               // wave-uniform: every lane takes part in the swaps, dead rows included
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        uint2 pk[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int g = 2 * gp + h;
                            const int ch = c0 + (wm * TM + i) * 32 + 8 * g + 4 * kh;
                            float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                            const bool on = ch < p.cout && live;
                            if (p.bias && on) {
                                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch);
                                v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                            }
                            if (yrow && on) {
                                const float4 y4 = use_pre ? El<T>::cvt4(ypre[(j * TM + i) * 4 + g]) : El<T>::ld4(yrow + ch);
                                v.x += y4.x; v.y += y4.y; v.z += y4.z; v.w += y4.w;
                            }
                            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                            const bf16x4 b = {(__bf16)activate(v.x, slope), (__bf16)activate(v.y, slope), (__bf16)activate(v.z, slope),
                                              (__bf16)activate(v.w, slope)};
                            pk[h] = __builtin_bit_cast(uint2, b);
                        }
                        const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        // lower lane: [own group 2gp | upper's group 2gp]; upper lane: [lower's group 2gp+1 | own group 2gp+1]
                        const int ch16 = c0 + (wm * TM + i) * 32 + 16 * gp + 8 * kh;
                        if (ch16 < p.cout && live)
                            *reinterpret_cast<uint4*>(orow + ch16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                    }
                }
                continue;
            }
        }
        if (vec && !(LSM && p.act == 3)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = c0 + (wm * TM + i) * 32 + 8 * g + 4 * kh;
                    float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    if (ch >= p.cout || !live) continue;
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    }
                    if (yrow) {
                        const float4 y4 = use_pre ? El<T>::cvt4(ypre[(j * TM + i) * 4 + g]) : El<T>::ld4(yrow + ch);
                        v.x += y4.x; v.y += y4.y; v.z += y4.z; v.w += y4.w;
                    }
                    El<T>::st4(orow + ch, make_float4(activate(v.x, slope), activate(v.y, slope), activate(v.z, slope),
                                                      activate(v.w, slope)));
                }
            }
        } else if (vec) {
            float4 v[TM][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = c0 + (wm * TM + i) * 32 + 8 * g + 4 * kh;
                    v[i][g] = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    if (ch >= p.cout || !live) continue;
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch);
                        v[i][g].x += b4.x; v[i][g].y += b4.y; v[i][g].z += b4.z; v[i][g].w += b4.w;
                    }
                    if (yrow) {
                        const float4 y4 = El<T>::ld4(yrow + ch);
                        v[i][g].x += y4.x; v[i][g].y += y4.y; v[i][g].z += y4.z; v[i][g].w += y4.w;
                    }
                }
            }
            {
                // log_softmax over the channels of a point (pspnet.py:108-112 `final`): the launcher guarantees that the
                // wave's tile spans all cout channels; a point's channels sit in lanes l and l ^ 32
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (i * 32 + 8 * g + 4 * kh < p.cout)
                            m = fmaxf(fmaxf(fmaxf(m, v[i][g].x), fmaxf(v[i][g].y, v[i][g].z)), v[i][g].w);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (i * 32 + 8 * g + 4 * kh < p.cout)
                            sum += (expf(v[i][g].x - m) + expf(v[i][g].y - m)) + (expf(v[i][g].z - m) + expf(v[i][g].w - m));
                sum += __shfl_xor(sum, 32, 64);
                const float lse = m + logf(sum);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = c0 + (wm * TM + i) * 32 + 8 * g + 4 * kh;
                        if (ch < p.cout && live)
                            El<T>::st4(orow + ch, make_float4(v[i][g].x - lse, v[i][g].y - lse, v[i][g].z - lse, v[i][g].w - lse));
                    }
            }
        } else if (live) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int ch = c0 + (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                    if (ch >= p.cout) continue;
                    float u = acc[i][j][q];
                    if (p.bias) u += p.bias[ch];
                    if (yrow) u += El<T>::ld(yrow + ch);
                    El<T>::st(orow + ch, activate(u, slope));
                }
            }
        }
    }
}

template <typename T, int TM, int TN, bool LSM = true>
__device__ __forceinline__ void pm_epilogue(const PmParams& p, f32x16 (&acc)[TM][TN], int c0, int r0, int wm, int wn, int l31, int kh)
{
    typename El<T>::Raw4 none[TM * TN * 4];           // never read
    pm_epilogue<T, TM, TN, LSM>(p, acc, c0, r0, wm, wn, l31, kh, none, false);
}

// csrc/mlp_pm_big.hip: the 256 x 256 bf16 tile with LDS-DMA operand loads; false = hipFuncSetAttribute failed
bool launch_pm_big_bf16(PmParams& p, hipStream_t st, int plan = 0);     // plan: tiles per workgroup | probe variant << 4 (csrc/mlp_pm_big.hip)
// ... can it run this launch?  (bf16 rows of whole 128-byte segments from each source, 16-byte aligned output / Y rows, no log-softmax)
bool big_form_ok(const PmParams& p);

}  // namespace pm
}  // namespace ffb6d
