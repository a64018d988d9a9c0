// ffb6d_amd/csrc/mfma_pm.h -- vocabulary shared by the point-major MFMA kernels (csrc/mlp_pm.hip, csrc/lfa_pm.hip):
// element-type traits of a row (float32 / bfloat16), one 32-byte k-step of a tile of 32 x 32 MFMAs, buffer descriptors.
//
//   v_mfma_f32_32x32x2_f32:   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; a lane holds ONE 16-byte chunk of
//   its row at k0 + 4*(l>>5) and the MFMA of sub-step t (t = 0..3) multiplies the pairs (k0+t, k0+4+t): 8 k per 32-byte step.
//   v_mfma_f32_32x32x16_bf16: 8 consecutive k per lane = the same 16-byte chunk: 16 k per step, one MFMA.
//   Accumulator register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31] in both.
#pragma once
#include <hip/hip_runtime.h>

namespace ffb6d {
namespace pm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BLK = 256;

// element-type traits: bytes, k per 32-byte step, 4-channel group load / store of the epilogue
template <typename T> struct El;
template <> struct El<float> {
    static constexpr int SZ = 4;
    typedef float4 Raw4;                                    // four channels as they are loaded (converted where they are used)
    static __device__ __forceinline__ Raw4 ld4raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ float4 cvt4(Raw4 r) { return r; }
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct El<__bf16> {
    static constexpr int SZ = 2;
    typedef uint2 Raw4;
    static __device__ __forceinline__ Raw4 ld4raw(const __bf16* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ float4 cvt4(Raw4 u)
    {
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ float4 ld4(const __bf16* p) { return cvt4(ld4raw(p)); }
    static __device__ __forceinline__ void st4(__bf16* p, float4 v)
    {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4*>(p) = b;
    }
    static __device__ __forceinline__ float ld(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ void st(__bf16* p, float v) { *p = (__bf16)v; }
};

// one 32-byte step of a TM x TN tile: fp32 = 8 k as four 32x32x2 MFMAs (pairs k0+t, k0+4+t), bf16 = 16 k as one 32x32x16
template <typename T, int TM, int TN>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[TM][TN], const u32x4 (&a)[TM], const u32x4 (&b)[TN])
{
    if constexpr (El<T>::SZ == 4) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][tt]), __uint_as_float(b[j][tt]),
                                                                     acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                    acc[i][j], 0, 0, 0);
    }
}

// act(v) = max(v, slope * v): slope 1 = identity, 0 = ReLU, 0.2 = LeakyReLU(0.2) -- no branch on the activation code
__device__ __forceinline__ float activate(float v, float slope) { return fmaxf(v, slope * v); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// LDS-DMA (`buffer_load_dwordx4 ... lds`): 16 bytes per lane from rs[voffset + soffset] straight into LDS at lds_dst + 16 * lane
// (lds_dst wave-uniform: the hardware takes it from M0; out-of-range lanes deliver zeros).  Counts on vmcnt like any buffer load;
// a reader in another wave needs the issuer's vmcnt wait AND a barrier.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_dst, int voffset, int soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

// wait until at most N of this wave's vector-memory operations are outstanding (they complete in order: the N youngest may remain)
template <int N>
__device__ __forceinline__ void lds_dma_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);      // vmcnt(N), expcnt / lgkmcnt left alone (gfx9 encoding)
#endif
}

// an integer the compiler cannot see through: what is computed from the result is computed HERE (not hoisted out of the enclosing loop,
// where it would hold registers across the whole loop)
__device__ __forceinline__ int opaque(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// make the compiler treat `v` as used HERE: the wait for the load that produces it is placed at this point of the instruction stream
// (and not after vector-memory operations issued later, whose completion that wait would then include)
__device__ __forceinline__ void use_here(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#else
    (void)v;
#endif
}

}  // namespace pm
}  // namespace ffb6d
