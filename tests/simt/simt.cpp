// tests/simt/simt.cpp -- fiber scheduler of the SIMT emulator (see simt.h).  Test infrastructure only.
#include "simt.h"

#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// Context switch of the fibers: callee-saved registers + stack pointer (System V x86-64).  ucontext would do, but its
// swapcontext makes a sigprocmask system call per switch and an emulated MFMA costs ~130 switches.
extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl simt_switch
    .type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size simt_switch, .-simt_switch
)");

namespace simt {

Dim3 g_thread, g_block, g_grid, g_bdim;

namespace {

constexpr size_t STACK = 512 << 10;
int WAVE = 64;                           // rendezvous width of the running launch (64 or 16)

struct Wave { int live = 0, arrived = 0; unsigned gen = 0; alignas(16) unsigned char scratch[64 * 256]; };
struct Fiber { void* sp = nullptr; bool done = false; int tid = 0; };

std::vector<Fiber> fibers;
std::vector<Wave> waves;
std::vector<unsigned char*> stacks;
void* sched_sp = nullptr;
int cur = -1;
int block_live = 0, block_arrived = 0;
unsigned block_gen = 0;
const std::function<void()>* body = nullptr;
alignas(16) unsigned char shared_mem[160 << 10];

void release_wave_if_complete(Wave& w)
{
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; ++w.gen; }
}
void release_block_if_complete()
{
    if (block_live > 0 && block_arrived == block_live) { block_arrived = 0; ++block_gen; }
}

void yield() { simt_switch(&fibers[cur].sp, sched_sp); }

void fiber_main()
{
    (*body)();
    Fiber& f = fibers[cur];
    f.done = true;
    Wave& w = waves[f.tid / WAVE];
    --w.live;
    --block_live;
    release_wave_if_complete(w);        // lanes that left no longer take part in rendezvous
    release_block_if_complete();
    simt_switch(&f.sp, sched_sp);
    abort();                            // a finished fiber is never resumed
}

}  // namespace

int lane() { return fibers[cur].tid % 64; }      // lane of the 64-wide wavefront, whatever the rendezvous width
int width() { return WAVE; }
unsigned char* dyn_shared() { return shared_mem; }
void* wave_scratch() { return waves[fibers[cur].tid / WAVE].scratch; }

void wave_sync()
{
    Wave& w = waves[fibers[cur].tid / WAVE];
    const unsigned gen = w.gen;
    ++w.arrived;
    release_wave_if_complete(w);
    while (w.gen == gen) yield();
}

void block_sync()
{
    const unsigned gen = block_gen;
    ++block_arrived;
    release_block_if_complete();
    while (block_gen == gen) yield();
}

void run_grid(Dim3 grid, Dim3 block, size_t dyn_shared_bytes, const std::function<void()>& thread_body, int width)
{
    WAVE = width;
    const int n = (int)(block.x * block.y * block.z);
    if (dyn_shared_bytes > sizeof(shared_mem)) { fprintf(stderr, "simt: %zu bytes of shared memory requested\n", dyn_shared_bytes); abort(); }
    while ((int)stacks.size() < n) {
        void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("simt: mmap"); abort(); }
        stacks.push_back(static_cast<unsigned char*>(p));
    }
    body = &thread_body;
    g_grid = grid;
    g_bdim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                fibers.assign(n, Fiber());
                waves.assign((n + WAVE - 1) / WAVE, Wave());
                block_live = n;
                block_arrived = 0;
                for (int t = 0; t < n; ++t) {
                    Fiber& f = fibers[t];
                    f.tid = t;
                    ++waves[t / WAVE].live;
                    // initial frame: six callee-saved registers, the entry point simt_switch "returns" to, and a slot
                    // that stands for the return address of a call (stack alignment at function entry)
                    void** top = reinterpret_cast<void**>(stacks[t] + STACK);
                    top[-1] = nullptr;
                    top[-2] = reinterpret_cast<void*>(&fiber_main);
                    for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
                    f.sp = top - 8;
                }
                int remaining = n;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < n; ++t) {
                        if (fibers[t].done) continue;
                        cur = t;
                        g_block = Dim3(bx, by, bz);
                        g_thread = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        simt_switch(&sched_sp, fibers[t].sp);
                        if (!fibers[t].done) ++remaining;
                    }
                }
            }
    body = nullptr;
    cur = -1;
}

}  // namespace simt
