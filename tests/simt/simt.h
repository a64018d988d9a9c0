// tests/simt/simt.h -- TEST INFRASTRUCTURE: a small SIMT emulator for running wave-level HIP kernels of the product
// library on the HOST (no GPU): every thread of a workgroup is a fiber (ucontext), workgroups run one after the other,
// cross-lane instructions (MFMA, shuffles), barriers and -- to keep wave-synchronous LDS idioms valid -- buffer loads /
// stores and sched_barrier are rendezvous points of the 64 fibers of a wave.  Functional only: no timing, no memory model
// beyond "a wave runs in lock step between rendezvous points".  Nothing in the product path uses this.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>

namespace simt {

struct Dim3 { unsigned x, y, z; Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

extern Dim3 g_thread, g_block, g_grid, g_bdim;      // threadIdx, blockIdx, gridDim, blockDim of the running fiber

// `width` = lanes that meet at a rendezvous: 64 (a wavefront), or 16 for kernels whose cross-lane traffic never leaves a DPP
// row and whose rows diverge from one another (a fiber model has no EXEC mask: lanes that skip a cross-lane instruction
// simply never arrive; with row-wide rendezvous the rows run independently, which is what such kernels compute anyway)
void run_grid(Dim3 grid, Dim3 block, size_t dyn_shared_bytes, const std::function<void()>& thread_body, int width = 64);
void wave_sync();                                   // rendezvous of the live lanes of the running fiber's wave
void block_sync();                                  // __syncthreads
int lane();                                         // lane of the running fiber in its wave
unsigned char* dyn_shared();                        // dynamic shared memory of the running workgroup (16-byte aligned)
void* wave_scratch();                               // 64 x 256 bytes exchanged through by the cross-lane emulations

int width();                                        // rendezvous width of the running launch

template <typename... A, typename... B>
void launch(const char* name, void (*kernel)(A...), Dim3 grid, Dim3 block, size_t shmem, B... args)
{
    const int w = strstr(name, "row16") ? 16 : 64;   // kernels named *row16*: one DPP row per query / point, rows independent
    run_grid(grid, block, shmem, [=]() { kernel(args...); }, w);
}

}  // namespace simt
