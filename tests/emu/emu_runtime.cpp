// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h): the fiber scheduler behind emu_launch /
// __syncthreads, and the few library-wide symbols that live in fec_engine.hip in the real library (error string, profiling
// scope, FEC entry points -- the twin covers the demodulator only, the FEC entries fail loudly).
#include "hip/hip_runtime.h"
#include "common.h"
#include "../../include/sdhip.h"
#include <cstdio>
#include <string>
#include <ucontext.h>
#include <vector>
#include <sys/mman.h>

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace
{
    constexpr size_t STACK = 256 * 1024;
    constexpr int MAXW = 16; // waves per block (1024 threads)
    struct Fiber
    {
        ucontext_t ctx;
        char *stack = nullptr;
        bool done = false;
    };
    std::vector<Fiber> fibers;
    ucontext_t sched_ctx;
    int current = -1;
    const std::function<void()> *body = nullptr;
    // barrier state of the running block: block-wide (__syncthreads) and per wave (collectives)
    int live_block = 0, barr_arrived = 0;
    unsigned bgen = 0;
    int live_wave[MAXW], w_arrived[MAXW];
    unsigned wgen[MAXW];
    unsigned long long xchg[MAXW][64];
    unsigned long long spins = 0;

    void yield() { swapcontext(&fibers[current].ctx, &sched_ctx); }
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
        swapcontext(&fibers[current].ctx, &sched_ctx);
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
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, fiber_main, 0);
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
                        swapcontext(&sched_ctx, &fibers[t].ctx);
                    }
                }
            }
    body = outer;
    current = -1;
}

// (error string, profiling scope and the C ABI's FEC half come from fec_engine.hip itself)
