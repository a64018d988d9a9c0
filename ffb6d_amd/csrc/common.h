// ffb6d_amd/csrc/common.h -- shared host-side helpers for the gfx950 library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "ffb6d_knn.h"

namespace ffb6d {

// thread-local last-error text surfaced through ffb6d_last_error()
char* err_buf();
int set_error(int code, const char* fmt, ...);

inline hipStream_t as_stream(ffb6d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;         // CDNA4 wavefront width
constexpr int kNumXCD = 8;        // MI355X: 8 XCDs, each with a private L2

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Kernel attributes (dynamic LDS limit) and occupancy answers belong to a DEVICE: a process that drives several GPUs must set /
// query them once per device, not once per process.  Slot of the calling thread's current device in a per-call-site cache
// (`static int cache[kMaxDevices]`, 0 = not initialised yet; racing initialisers write the same value).
constexpr int kMaxDevices = 64;
inline int device_slot()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

}  // namespace ffb6d

#define FFB6D_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return ffb6d::set_error(FFB6D_ERR_HIP, "%s failed: %s (%s:%d)", #expr,       \
                                    hipGetErrorString(e__), __FILE__, __LINE__);         \
    } while (0)

#define FFB6D_LAUNCH_CHECK()                                                             \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess)                                                           \
            return ffb6d::set_error(FFB6D_ERR_HIP, "kernel launch failed: %s (%s:%d)",   \
                                    hipGetErrorString(e__), __FILE__, __LINE__);         \
    } while (0)

#define FFB6D_REQUIRE(cond, ...)                                                         \
    do {                                                                                 \
        if (!(cond)) return ffb6d::set_error(FFB6D_ERR_ARG, __VA_ARGS__);                \
    } while (0)
