// tests/simt/fake/hip/hip_runtime.h -- TEST INFRASTRUCTURE: the slice of the HIP device / runtime vocabulary that the
// product kernels use, re-defined for the HOST on top of the SIMT emulator (tests/simt/simt.h), so that the unmodified kernel
// sources compile with the host compiler (clang++ -x c++) and run on the CPU.  Only what csrc/*.hip needs is here.
//
// gfx950 semantics restated:
//   v_mfma_f32_32x32x2_f32   lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; accumulator register r of
//                            lane l is D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
//   v_mfma_f32_32x32x16_bf16 same accumulator layout; lane l supplies A[i = l & 31][k = 8 (l >> 5) .. + 7] and the same of B
//   raw buffer load / store  per dword: in range iff voffset + 4 d < num_records (the scalar offset is not range checked);
//                            out-of-range loads return zero, stores are dropped
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "simt.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static                      /* one workgroup runs at a time: a static array IS workgroup-shared */
#define __restrict__

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef simt::Dim3 dim3;
#define threadIdx simt::g_thread
#define blockIdx simt::g_block
#define gridDim simt::g_grid
#define blockDim simt::g_bdim

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "simt"; }
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s_, size_t n, hipMemcpyKind) { memcpy(d, s_, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s_, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n) { memcpy(sym, src, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 4; return hipSuccess; }   /* a small "chip": persistent kernels walk several tiles */
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) simt::launch(#kernel, kernel, grid, block, shmem, __VA_ARGS__)

static inline float __uint_as_float(unsigned u) { return __builtin_bit_cast(float, u); }
static inline unsigned __float_as_uint(float f) { return __builtin_bit_cast(unsigned, f); }
static inline float __int_as_float(int i) { return __builtin_bit_cast(float, i); }
static inline int __float_as_int(float f) { return __builtin_bit_cast(int, f); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
template <typename T> static inline T atomicMax(T* p, T v) { const T old = *p; if (v > old) *p = v; return old; }
template <typename T> static inline T atomicMin(T* p, T v) { const T old = *p; if (v < old) *p = v; return old; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned old = *p; *p = old + v; return old; }
static inline float atomicAdd(float* p, float v) { const float old = *p; *p = old + v; return old; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long old = *p; *p = old + v; return old; }
static inline int atomicAdd(int* p, int v) { const int old = *p; *p = old + v; return old; }   /* fibers never run concurrently */
static inline float unsafeAtomicAdd(float* p, float v) { const float old = *p; *p = old + v; return old; }
static inline void __syncthreads() { simt::block_sync(); }
static inline int min(int a, int b) { return a < b ? a : b; }                 /* HIP's device-side overloads */
static inline int max(int a, int b) { return a > b ? a : b; }
/* round-to-nearest arithmetic that the compiler must not contract (this library is built with -ffp-contract=off) */
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }

namespace simt {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct Rsrc { unsigned char* base; unsigned size; };

template <typename V> inline V* scratch() { return static_cast<V*>(wave_scratch()); }

// __shfl_xor: value of lane l ^ mask; overloaded like HIP's (an int must not travel as a float)
inline unsigned long long shfl_xor(unsigned long long v, int mask)
{
    unsigned long long* s = scratch<unsigned long long>();
    s[lane()] = v;
    wave_sync();
    const unsigned long long r = s[(lane() ^ mask) & 63];
    wave_sync();
    return r;
}
inline long long shfl_xor(long long v, int mask) { return (long long)shfl_xor((unsigned long long)v, mask); }
inline int shfl_xor(int v, int mask) { return (int)shfl_xor((unsigned long long)(unsigned)v, mask); }
inline unsigned shfl_xor(unsigned v, int mask) { return (unsigned)shfl_xor((unsigned long long)v, mask); }
inline float shfl_xor(float v, int mask) { return __builtin_bit_cast(float, shfl_xor(__builtin_bit_cast(unsigned, v), mask)); }

// v_mov_b32_dpp with a row control (rows of 16 lanes): row_shl:n 0x100+n, row_shr:n 0x110+n, row_ror:n 0x120+n.
// Lane i of a row receives lane i+n (shl), i-n (shr) or (i-n) mod 16 (ror) of the same row; a lane whose source falls off
// the row keeps `old` (bound_ctrl = false) or gets 0 (true).  row_mask / bank_mask 0xf only.
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    int* s = scratch<int>();
    const int l = lane();
    s[l] = src;
    wave_sync();
    const int i = l & 15, n = ctrl & 15, kind = ctrl & 0x1f0;
    int from = -1;
    if (kind == 0x100) from = i + n;
    else if (kind == 0x110) from = i - n;
    else if (kind == 0x120) from = (i - n) & 15;
    else { fprintf(stderr, "simt: DPP control 0x%x is not emulated\n", ctrl); abort(); }
    if (row_mask != 0xf || bank_mask != 0xf) { fprintf(stderr, "simt: DPP row/bank masks are not emulated\n"); abort(); }
    const int r = (from >= 0 && from < 16) ? s[(l & ~15) + from] : (bound_ctrl ? 0 : old);
    wave_sync();
    return r;
}

// lanes [base, base + width) meet; with a 16-lane rendezvous the other rows' bits read as 0
inline unsigned long long ballot(int pred)
{
    int* s = scratch<int>();
    s[lane()] = pred ? 1 : 0;
    wave_sync();
    const int base = lane() & ~(width() - 1);
    unsigned long long m = 0;
    for (int l = base; l < base + width(); ++l) m |= (unsigned long long)(s[l] & 1) << l;
    wave_sync();
    return m;
}

// __shfl(v, src, w): value of lane (l & ~(w-1)) + src % w
inline int shfl(int v, int src, int w)
{
    int* s = scratch<int>();
    const int l = lane();
    s[l] = v;
    wave_sync();
    const int r = s[(l & ~(w - 1)) + (src & (w - 1))];
    wave_sync();
    return r;
}
// __shfl_up(v, d, w): value of lane l - d of the same w-wide group; lanes whose source falls off keep their own value
inline int shfl_up(int v, int d, int w)
{
    int* s = scratch<int>();
    const int l = lane();
    s[l] = v;
    wave_sync();
    const int r = ((l & (w - 1)) >= d) ? s[l - d] : v;
    wave_sync();
    return r;
}
inline float shfl(float v, int src, int w) { return __builtin_bit_cast(float, shfl(__builtin_bit_cast(int, v), src, w)); }
inline unsigned shfl(unsigned v, int src, int w) { return (unsigned)shfl((int)v, src, w); }
// v_permlane32_swap_b32 vdst, vsrc: lanes 32-63 of vdst trade places with lanes 0-31 of vsrc; returns {vdst', vsrc'}
typedef unsigned u32x2_swap_t __attribute__((ext_vector_type(2)));
inline u32x2_swap_t permlane32_swap(unsigned vdst, unsigned vsrc)
{
    const int l = lane();
    const unsigned dst_up = shfl(vdst, (l + 32) & 63, 64);       // vdst of the lane 32 above (read by the lower half)
    const unsigned src_lo = shfl(vsrc, (l + 32) & 63, 64);       // vsrc of the lane 32 below (read by the upper half)
    u32x2_swap_t r;
    r[0] = l < 32 ? vdst : src_lo;
    r[1] = l < 32 ? dst_up : vsrc;
    return r;
}

inline f32x16_t mfma_32x32x2_f32(float a, float b, f32x16_t c)
{
    struct Op { float a, b; };
    Op* s = scratch<Op>();
    const int l = lane();
    s[l] = Op{a, b};
    wave_sync();
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(s[i + 32 * k].a, s[j + 32 * k].b, acc);
        c[r] = acc;
    }
    wave_sync();
    return c;
}

inline f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c)
{
    struct Op { float a[8], b[8]; };
    Op* s = scratch<Op>();
    const int l = lane();
    for (int e = 0; e < 8; ++e) { s[l].a[e] = (float)a[e]; s[l].b[e] = (float)b[e]; }
    wave_sync();
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) acc = fmaf(s[i + 32 * h].a[e], s[j + 32 * h].b[e], acc);
        c[r] = acc;
    }
    wave_sync();
    return c;
}

inline u32x4_t buffer_load_b128(Rsrc rs, int voffset, int soffset)
{
    wave_sync();                                     // a wave issues it in lock step (wave-synchronous LDS idioms rely on that)
    u32x4_t v = {0u, 0u, 0u, 0u};
    const unsigned vo = (unsigned)voffset;
    for (int d = 0; d < 4; ++d) {
        const unsigned off = vo + 4u * d;
        if (off < rs.size && vo <= 0xffffffffu - 16u) {
            unsigned w;
            memcpy(&w, rs.base + (ptrdiff_t)soffset + off, 4);
            v[d] = w;
        }
    }
    return v;
}

// buffer_load_dwordx4 ... lds: 16 bytes per lane, range checked like the register form, written to lds_base + 16 * lane
// (lds_base wave-uniform: M0 on the hardware)
inline void buffer_load_lds16(Rsrc rs, unsigned char* lds_base, int voffset, int soffset)
{
    const u32x4_t v = buffer_load_b128(rs, voffset, soffset);
    memcpy(lds_base + 16 * lane(), &v, 16);
    wave_sync();
}

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
inline u32x2_t buffer_load_b64(Rsrc rs, int voffset, int soffset)
{
    wave_sync();
    u32x2_t v = {0u, 0u};
    const unsigned vo = (unsigned)voffset;
    for (int d = 0; d < 2; ++d) {
        const unsigned off = vo + 4u * d;
        if (off < rs.size && vo <= 0xffffffffu - 8u) {
            unsigned w;
            memcpy(&w, rs.base + (ptrdiff_t)soffset + off, 4);
            v[d] = w;
        }
    }
    return v;
}

inline void buffer_store_b128(u32x4_t v, Rsrc rs, int voffset, int soffset)
{
    wave_sync();
    const unsigned vo = (unsigned)voffset;
    for (int d = 0; d < 4; ++d) {
        const unsigned off = vo + 4u * d;
        if (off < rs.size && vo <= 0xffffffffu - 16u) {
            const unsigned w = v[d];
            memcpy(rs.base + (ptrdiff_t)soffset + off, &w, 4);
        }
    }
}

}  // namespace simt

typedef simt::Rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(ptr, stride, num, flags) simt::Rsrc{reinterpret_cast<unsigned char*>(ptr), (unsigned)(num)}
#define __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, aux) simt::buffer_load_b128(rs, vo, so)
#define __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, aux) simt::buffer_load_b64(rs, vo, so)
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ldsp, size, vo, so, imm, aux) simt::buffer_load_lds16(rs, (unsigned char*)(ldsp), vo, so)
#define __builtin_amdgcn_raw_buffer_store_b128(v, rs, vo, so, aux) simt::buffer_store_b128(v, rs, vo, so)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) simt::mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) simt::mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_readfirstlane(x) (x)   /* the kernels only pass wave-uniform values */
#define __builtin_amdgcn_sched_barrier(x) simt::wave_sync()
// scheduling hints and hardware-id reads: no functional effect on the emulator (workgroups run one after the other)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)             /* memory operations complete at once on the emulator */
#define __builtin_amdgcn_s_barrier() simt::block_sync()
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) simt::permlane32_swap(a, b)
#define __builtin_amdgcn_s_getreg(reg_) ((((reg_) & 63) == 20) ? (unsigned)(blockIdx.x & 7u) : 0u)      /* HW_REG_XCC_ID: workgroup b on XCD b % 8 */
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __shfl_xor(v, mask, ...) simt::shfl_xor(v, mask)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) simt::update_dpp(old, src, ctrl, rm, bm, bc)
#define __ballot(p) simt::ballot(p)
#define __any(p) (simt::ballot(p) != 0)
#define __shfl(v, src, w) simt::shfl(v, src, w)
#define __shfl_up(v, d, w) simt::shfl_up(v, d, w)
#define __threadfence_block() ((void)0)
#define __builtin_amdgcn_readlane(v, l) simt::shfl((int)(v), l, 64)
#define __builtin_amdgcn_wave_barrier() simt::wave_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
