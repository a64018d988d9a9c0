// ffb6d_amd/csrc/row_unit.h -- a 16-byte unit of a point-major / pixel-major row: VL consecutive channels (4 float32 or
// 8 bfloat16), held as fp32 in registers; stores round to nearest even.  __host__ __device__: the per-thread kernel bodies
// that use it (csrc/*_body.h) also run on the CPU in tests/hostsim.
#pragma once
#include <hip/hip_runtime.h>

namespace ffb6d {

template <typename T> struct RowUnit;
template <> struct RowUnit<float> {
    static constexpr int VL = 4;
    float v[4];
    static __host__ __device__ __forceinline__ RowUnit load(const void* base, size_t unit)
    {
        const float4 f = static_cast<const float4*>(base)[unit];
        RowUnit u; u.v[0] = f.x; u.v[1] = f.y; u.v[2] = f.z; u.v[3] = f.w;
        return u;
    }
    static __host__ __device__ __forceinline__ RowUnit unpack(uint4 w)         // a unit fetched earlier as raw bits
    {
        RowUnit u;
        u.v[0] = __builtin_bit_cast(float, w.x); u.v[1] = __builtin_bit_cast(float, w.y);
        u.v[2] = __builtin_bit_cast(float, w.z); u.v[3] = __builtin_bit_cast(float, w.w);
        return u;
    }
    __host__ __device__ __forceinline__ void store(void* base, size_t unit) const
    {
        static_cast<float4*>(base)[unit] = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct RowUnit<__bf16> {
    static constexpr int VL = 8;
    float v[8];
    static __host__ __device__ __forceinline__ RowUnit load(const void* base, size_t unit)
    {
        return unpack(static_cast<const uint4*>(base)[unit]);
    }
    static __host__ __device__ __forceinline__ RowUnit unpack(uint4 w)
    {
        const unsigned int x[4] = {w.x, w.y, w.z, w.w};
        RowUnit u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u.v[2 * i] = __builtin_bit_cast(float, x[i] << 16);
            u.v[2 * i + 1] = __builtin_bit_cast(float, x[i] & 0xffff0000u);
        }
        return u;
    }
    __host__ __device__ __forceinline__ void store(void* base, size_t unit) const
    {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (__bf16)v[i];       // round to nearest even
        static_cast<bf16x8*>(base)[unit] = b;
    }
};

// half a bf16 unit: 4 consecutive channels in 8 bytes -- the register footprint of a float32 unit, for per-thread bodies whose
// register blocking does not fit with 8-channel units (csrc/upconv_body.h: the 2 x 4 output block)
struct Bf16Half {};
template <> struct RowUnit<Bf16Half> {
    static constexpr int VL = 4;
    float v[4];
    static __host__ __device__ __forceinline__ RowUnit load(const void* base, size_t unit)
    {
        const uint2 w = static_cast<const uint2*>(base)[unit];
        RowUnit u;
        u.v[0] = __builtin_bit_cast(float, w.x << 16); u.v[1] = __builtin_bit_cast(float, w.x & 0xffff0000u);
        u.v[2] = __builtin_bit_cast(float, w.y << 16); u.v[3] = __builtin_bit_cast(float, w.y & 0xffff0000u);
        return u;
    }
    __host__ __device__ __forceinline__ void store(void* base, size_t unit) const
    {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};       // round to nearest even
        static_cast<bf16x4*>(base)[unit] = b;
    }
};

}  // namespace ffb6d
