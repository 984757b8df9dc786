// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h): the fiber scheduler behind emu_launch /
// __syncthreads, and the few library-wide symbols that live in fec_engine.hip in the real library (error string, profiling
// scope, FEC entry points -- the twin covers the demodulator only, the FEC entries fail loudly).
#include "hip/hip_runtime.h"
#include "common.h"
#include "../../include/sdhip.h"
#include <cstdio>
#include <string>
#include <cstdint>
#include <cstring>
#include <vector>
#include <sys/mman.h>

// Fiber switch. ucontext's swapcontext makes a signal-mask system call per switch, and a wave collective is 256 switches: the FEC half of the twin spent its time
// there. This one saves what the System V x86-64 ABI calls callee-saved (rbx, rbp, r12-r15, the SSE / x87 control words) and changes stacks: a few nanoseconds.
#if !defined(__x86_64__)
#error "tests/emu: the fiber switch is written for x86-64 hosts"
#endif
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch,.-emu_switch
)");

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace
{
    constexpr size_t STACK = 256 * 1024;
    constexpr int MAXW = 16; // waves per block (1024 threads)
    struct Fiber
    {
        void *sp = nullptr;
        char *stack = nullptr;
        bool done = false;
    };
    std::vector<Fiber> fibers;
    void *sched_sp = nullptr;
    int current = -1;
    const std::function<void()> *body = nullptr;
    // barrier state of the running block: block-wide (__syncthreads) and per wave (collectives)
    int live_block = 0, barr_arrived = 0;
    unsigned bgen = 0;
    int live_wave[MAXW], w_arrived[MAXW];
    unsigned wgen[MAXW];
    unsigned long long xchg[MAXW][64];
    unsigned long long spins = 0;

    void yield() { emu_switch(&fibers[current].sp, sched_sp); }
    void release_if_complete(int w)
    { // called when an arrival OR an exit may have completed a rendezvous
        if (live_block > 0 && barr_arrived == live_block)
        {
            barr_arrived = 0;
            bgen++;
        }
        if (w >= 0 && live_wave[w] > 0 && w_arrived[w] == live_wave[w])
        {
            w_arrived[w] = 0;
            wgen[w]++;
        }
    }
    void fiber_main()
    {
        (*body)();
        fibers[current].done = true;
        live_block--;
        live_wave[current / 64]--;
        release_if_complete(current / 64); // threads that have left do not take part in later barriers (s_barrier semantics)
        emu_switch(&fibers[current].sp, sched_sp);
        abort(); // a finished fiber is never resumed
    }
    void fiber_prepare(Fiber &f)
    { // the stack emu_switch pops when it first switches to the fiber: control words, six registers, fiber_main as the return address (rsp = 8 mod 16 on entry)
        uint64_t *top = (uint64_t *)(f.stack + STACK);
        top[-1] = 0;
        top[-2] = (uint64_t)(uintptr_t)&fiber_main;
        for (int i = 3; i <= 8; i++)
            top[-i] = 0;
        const uint32_t csr[2] = {0x1F80u, 0x037Fu};
        memcpy(&top[-9], csr, 8);
        f.sp = &top[-9];
    }
    void wave_sync()
    {
        const int w = current / 64;
        const unsigned g = wgen[w];
        w_arrived[w]++;
        release_if_complete(w);
        while (wgen[w] == g)
        {
            if (++spins > 400000000ull)
            {
                fprintf(stderr, "host twin: wave collective never completed (divergent call?)\n");
                abort();
            }
            yield();
        }
        spins = 0;
    }
}

void __syncthreads()
{
    const unsigned g = bgen;
    barr_arrived++;
    release_if_complete(-1);
    while (bgen == g)
        yield();
}
int __syncthreads_or(int pred)
{ // block-wide OR: contributions, barrier, everybody reads, barrier, everybody clears, barrier (nobody runs ahead into the next call)
    static int acc = 0;
    if (pred)
        acc = 1;
    __syncthreads();
    const int r = acc;
    __syncthreads();
    acc = 0;
    __syncthreads();
    return r;
}
unsigned long long emu_wave_xchg(unsigned long long v, int src_lane)
{
    const int w = current / 64;
    xchg[w][current % 64] = v;
    wave_sync();
    const unsigned long long r = xchg[w][src_lane & 63];
    wave_sync(); // nobody overwrites a slot before everybody has read
    return r;
}
unsigned long long emu_ballot(int pred)
{
    const int w = current / 64;
    xchg[w][current % 64] = pred ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++)
    {
        const size_t t = (size_t)w * 64 + l;
        if (t < fibers.size() && t < (size_t)(blockDim.x * blockDim.y * blockDim.z) && !fibers[t].done && xchg[w][l])
            m |= 1ull << l;
    }
    wave_sync();
    return m;
}

void emu_launch(dim3 grid, dim3 block, const std::function<void()> &thread_body)
{
    const unsigned nt = block.x * block.y * block.z;
    if (nt > 64 * MAXW)
        abort();
    if (fibers.size() < nt)
        fibers.resize(nt);
    for (unsigned t = 0; t < nt; t++)
        if (!fibers[t].stack)
        {
            fibers[t].stack = (char *)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (fibers[t].stack == (char *)MAP_FAILED)
                abort();
        }
    const std::function<void()> *outer = body; // (kernels do not launch kernels; kept simple)
    body = &thread_body;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++)
            {
                live_block = (int)nt;
                barr_arrived = 0;
                for (int w = 0; w < MAXW; w++)
                {
                    const int lo = w * 64, hi = std::min<int>((w + 1) * 64, (int)nt);
                    live_wave[w] = hi > lo ? hi - lo : 0;
                    w_arrived[w] = 0;
                }
                for (unsigned t = 0; t < nt; t++)
                {
                    Fiber &f = fibers[t];
                    f.done = false;
                    fiber_prepare(f);
                }
                while (live_block > 0)
                { // round robin: every live fiber runs until it blocks in a rendezvous (it re-checks when resumed) or ends
                    for (unsigned t = 0; t < nt; t++)
                    {
                        if (fibers[t].done)
                            continue;
                        blockIdx = emu_idx{bx, by, bz};
                        threadIdx = emu_idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                        current = (int)t;
                        emu_switch(&sched_sp, fibers[t].sp);
                    }
                }
            }
    body = outer;
    current = -1;
}

// (error string, profiling scope and the C ABI's FEC half come from fec_engine.hip itself)
