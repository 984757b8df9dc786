// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h): the fiber scheduler behind emu_launch /
// __syncthreads, and the few library-wide symbols that live in fec_engine.hip in the real library (error string, profiling
// scope, FEC entry points -- the twin covers the demodulator only, the FEC entries fail loudly).
#include "hip/hip_runtime.h"
#include "common.h"
#include "../../include/sdhip.h"
#include <string>
#include <ucontext.h>
#include <vector>
#include <sys/mman.h>

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace
{
    constexpr size_t STACK = 256 * 1024;
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

    void fiber_main()
    {
        (*body)();
        fibers[current].done = true;
        swapcontext(&fibers[current].ctx, &sched_ctx);
    }
}

void __syncthreads()
{ // back to the scheduler; it resumes this fiber once every live fiber of the block has arrived (or finished)
    swapcontext(&fibers[current].ctx, &sched_ctx);
}

void emu_launch(dim3 grid, dim3 block, const std::function<void()> &thread_body)
{
    const unsigned nt = block.x * block.y * block.z;
    if (fibers.size() < nt)
        fibers.resize(nt);
    for (unsigned t = 0; t < nt; t++)
        if (!fibers[t].stack)
        {
            fibers[t].stack = (char *)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (fibers[t].stack == (char *)MAP_FAILED)
                abort();
        }
    body = &thread_body;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++)
            {
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
                unsigned live = nt;
                while (live)
                { // one pass = every live fiber runs up to its next barrier (or to its end)
                    for (unsigned t = 0; t < nt; t++)
                    {
                        if (fibers[t].done)
                            continue;
                        blockIdx = emu_idx{bx, by, bz};
                        threadIdx = emu_idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                        current = (int)t;
                        swapcontext(&sched_ctx, &fibers[t].ctx);
                        if (fibers[t].done)
                            live--;
                    }
                }
            }
    body = nullptr;
    current = -1;
}

namespace sdhip
{
    static std::string g_last_error;
    void set_error(const std::string &msg) { g_last_error = msg; }
    ProfScope::ProfScope(const char *, hipStream_t stream) : idx(-1), st(stream) {}
    ProfScope::~ProfScope() {}
}

extern "C"
{
    const char *sdhip_last_error(void) { return sdhip::g_last_error.c_str(); }
    const char *sdhip_version(void) { return "sdhip host twin (tests only)"; }
    void sdhip_prof_enable(int) {}
    void sdhip_prof_reset(void) {}
    int sdhip_prof_get(int, char *, size_t, double *, long long *) { return 0; }
    static void *no_fec(void)
    {
        sdhip::set_error("the host twin covers the demodulator only");
        return nullptr;
    }
    void sdhip_fec_cfg_default(sdhip_fec_cfg *c) { memset(c, 0, sizeof(*c)); }
    void *sdhip_fec_create(const sdhip_fec_cfg *) { return no_fec(); }
    void sdhip_fec_destroy(void *) {}
    int sdhip_fec_push(void *, const int8_t *, size_t) { return no_fec(), -1; }
    int64_t sdhip_fec_pull(void *, uint8_t *, size_t) { return no_fec(), -1; }
    int64_t sdhip_fec_process_dev(void *, const int8_t *, size_t, uint8_t *, size_t) { return no_fec(), -1; }
    int sdhip_fec_get_stats(void *, sdhip_fec_stats *) { return no_fec(), -1; }
    int64_t sdhip_fec_get_block_taps(void *, float *, int *, size_t) { return no_fec(), -1; }
    int sdhip_op_ccdecoder(int, int, const uint8_t *, int, uint8_t *) { return no_fec(), -1; }
    int sdhip_op_rs_decode(int, uint8_t *, int, int, int, int, int, int, int *) { return no_fec(), -1; }
}
