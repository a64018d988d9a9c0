// tests/simt/simt.h -- TEST INFRASTRUCTURE: a small SIMT emulator for running wave-level HIP kernels of the product
// library on the HOST (no GPU): every thread of a workgroup is a fiber (ucontext), workgroups run one after the other,
// cross-lane instructions (MFMA, shuffles), barriers and -- to keep wave-synchronous LDS idioms valid -- buffer loads /
// stores and sched_barrier are rendezvous points of the 64 fibers of a wave.  Functional only: no timing, no memory model
// beyond "a wave runs in lock step between rendezvous points".  Nothing in the product path uses this.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

namespace simt {

struct Dim3 { unsigned x, y, z; Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

extern Dim3 g_thread, g_block, g_grid, g_bdim;      // threadIdx, blockIdx, gridDim, blockDim of the running fiber

void run_grid(Dim3 grid, Dim3 block, size_t dyn_shared_bytes, const std::function<void()>& thread_body);
void wave_sync();                                   // rendezvous of the live lanes of the running fiber's wave
void block_sync();                                  // __syncthreads
int lane();                                         // lane of the running fiber in its wave
unsigned char* dyn_shared();                        // dynamic shared memory of the running workgroup (16-byte aligned)
void* wave_scratch();                               // 64 x 256 bytes exchanged through by the cross-lane emulations

template <typename... A, typename... B>
void launch(void (*kernel)(A...), Dim3 grid, Dim3 block, size_t shmem, B... args)
{
    run_grid(grid, block, shmem, [=]() { kernel(args...); });
}

}  // namespace simt
