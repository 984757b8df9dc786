// oracle/ref_wrap_ndsp.cpp -- TEST INFRASTRUCTURE (never shipped, never on the product path).
// Runs the REFERENCE's own ndsp blocks (src-core/dsp/**, compiled in place by oracle/Makefile into _ref/libsdref_ndsp.so) the way the
// reference runs them: each block on its own thread, fed through DSPStream FIFOs (src-core/dsp/base/stream.h), terminated by a propagating
// terminator buffer. One generic entry point: the block is chosen by its reference id, configured through the block's own set_cfg()
// (same keys the flowgraph uses), so nothing here restates any arithmetic.
#include "dsp/agc/agc.h"
#include "dsp/agc/agc_fast.h"
#include "dsp/clock_recovery/clock_recovery_gardner.h"
#include "dsp/clock_recovery/clock_recovery_mm.h"
#include "dsp/clock_recovery/clock_recovery_mm_fast.h"
#include "dsp/filter/fir.h"
#include "dsp/filter/rrc.h"
#include "dsp/hier/psk_demod.h"
#include "dsp/pll/costas.h"
#include "dsp/pll/costas_fast.h"
#include <cstring>
#include <memory>
#include <thread>

using namespace satdump::ndsp;

namespace
{
    std::unique_ptr<Block> make_block(const std::string &id)
    {
        if (id == "agc_cc")
            return std::make_unique<AGCBlock<complex_t>>();
        if (id == "agc_fast_cc")
            return std::make_unique<AGCFastBlock<complex_t>>();
        if (id == "rrc_fir_cc")
            return std::make_unique<RRC_Block<FIRBlock<complex_t>>>();
        if (id == "costas_cc")
            return std::make_unique<CostasBlock>();
        if (id == "costas_fast_cc")
            return std::make_unique<CostasFastBlock>();
        if (id == "fast_clock_recovery_mm_cc")
            return std::make_unique<MMClockRecoveryFastBlock<complex_t>>();
        if (id == "clock_recovery_mm_cc")
            return std::make_unique<MMClockRecoveryBlock<complex_t>>();
        if (id == "clock_recovery_gardner_cc")
            return std::make_unique<GardnerClockRecoveryBlock<complex_t>>();
        if (id == "psk_demod_cc")
            return std::make_unique<PSKDemodHierBlock>();
        return nullptr;
    }
} // namespace

extern "C"
{
    // in: n complex samples, handed to the block in buffers of `buf` samples. Returns the number of complex samples written to out
    // (-1: unknown block / bad cfg, -2: out too small). cfg_json: {"key": value, ...} applied in order with set_cfg(key, value).
    long long sdref_ndsp_run(const char *block_id, const char *cfg_json, const float *in, size_t n, size_t buf, float *out, size_t cap)
    {
        try
        {
            std::unique_ptr<Block> blk = make_block(block_id);
            if (!blk)
                return -1;
            nlohmann::ordered_json cfg = nlohmann::ordered_json::parse(cfg_json);
            for (auto &kv : cfg.items())
                if (blk->set_cfg(kv.key(), nlohmann::json(kv.value())) == Block::RES_ERR)
                    return -1;

            BlockIO src{"in", DSP_SAMPLE_TYPE_CF32};
            src.fifo = std::make_shared<DSPStream>(4);
            blk->set_input(src, 0);
            const bool hier = std::string(block_id) == "psk_demod_cc";
            BlockIO dst = blk->get_output(0, hier ? 0 : 4); // the hier block's output FIFO belongs to its splitter
            blk->start();

            std::thread feeder(
                [&]()
                {
                    for (size_t o = 0; o < n; o += buf)
                    {
                        const size_t m = std::min(buf, n - o);
                        DSPBuffer b = src.fifo->newBufferSamples((uint32_t)buf, sizeof(complex_t));
                        memcpy(b.getSamples<complex_t>(), in + 2 * o, m * sizeof(complex_t));
                        b.size = (uint32_t)m;
                        src.fifo->wait_enqueue(b);
                    }
                    src.fifo->wait_enqueue(src.fifo->newBufferTerminator());
                });

            long long got = 0;
            bool overflow = false;
            for (;;)
            {
                DSPBuffer b = dst.fifo->wait_dequeue();
                if (b.isTerminator())
                {
                    dst.fifo->free(b);
                    break;
                }
                if ((size_t)got + b.size > cap)
                    overflow = true;
                else
                {
                    memcpy(out + 2 * got, b.getSamples<complex_t>(), b.size * sizeof(complex_t));
                    got += b.size;
                }
                dst.fifo->free(b);
            }
            feeder.join();
            blk->stop();
            return overflow ? -2 : got;
        }
        catch (std::exception &e)
        {
            fprintf(stderr, "sdref_ndsp_run: %s\n", e.what());
            return -1;
        }
    }
}
