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
// (`static int cache[kMaxDevices + 1]`, 0 = not initialised yet; racing initialisers write the same value).  A device the cache has no
// slot for (ordinal >= kMaxDevices, or hipGetDevice failing) gets the slot kNoDeviceSlot, which call sites NEVER mark as initialised:
// they then set the attribute / ask the occupancy on every call instead of borrowing device 0's answer.
constexpr int kMaxDevices = 64;
constexpr int kNoDeviceSlot = kMaxDevices;
inline int device_slot()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return kNoDeviceSlot;
    return d;
}
// caches at the call sites: `static int flag[kMaxDevices + 1]`; cache_get / cache_set leave the overflow slot alone
inline int cache_get(const int* cache, int slot) { return slot == kNoDeviceSlot ? 0 : __atomic_load_n(&cache[slot], __ATOMIC_RELAXED); }
inline void cache_set(int* cache, int slot, int v) { if (slot != kNoDeviceSlot) __atomic_store_n(&cache[slot], v, __ATOMIC_RELAXED); }

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
