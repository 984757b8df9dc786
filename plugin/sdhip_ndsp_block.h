// sdhip_ndsp_block.h -- the MI355X PSK demodulator behind the reference's NEW block API (satdump::ndsp::Block, src-core/dsp/block.h:102-397).
//
// PSKDemodHipBlock is a drop-in for satdump::ndsp::PSKDemodHierBlock (src-core/dsp/hier/psk_demod.h): one cf32 input, one cf32 output,
// the same set_cfg()/get_cfg() keys, defaults and result codes; the member blocks' threads and FIFOs in between (RRC -> AGC -> M&M ->
// Costas) are replaced by one work() that hands every DSPBuffer to the C ABI (include/sdhip.h, sdhip_ndsp_psk_demod_*). Two keys the
// reference block does not have: "device" (GPU index) and "exact" (bit-exact sequential lanes instead of the chunk-parallel schedule).
// The reference block's "snr" statistic (its internal splitter + PSKSnrEstimatorBlock) is not computed here: connect the reference's own
// PSKSnrEstimatorBlock to the output if it is wanted.
#pragma once

#include "core/exception.h"
#include "dsp/block.h"
#include "common/dsp/block.h" // dsp::rad_to_hz
#include "common/dsp/complex.h"

#include "../include/sdhip.h"

#include <string>

namespace sdhip_plugin
{
    class PSKDemodHipBlock : public satdump::ndsp::Block
    {
    private:
        sdhip_ndsp_psk_cfg cfg;
        void *h = nullptr;
        std::string constellation = "bpsk";
        bool advanced_mode = false;
        bool needs_reinit = true;

        void drop()
        {
            if (h)
                sdhip_ndsp_psk_demod_destroy(h);
            h = nullptr;
        }

        bool work()
        {
            using namespace satdump::ndsp;
            DSPBuffer iblk = inputs[0].fifo->wait_dequeue();
            if (iblk.isTerminator())
            { // BlockSimple::work, block_simple.h:31-37
                if (iblk.terminatorShouldPropagate())
                    outputs[0].fifo->wait_enqueue(outputs[0].fifo->newBufferTerminator());
                inputs[0].fifo->free(iblk);
                return true;
            }
            if (needs_reinit)
            {
                needs_reinit = false;
                init();
            }
            DSPBuffer oblk = outputs[0].fifo->newBufferSamples(iblk.max_size, sizeof(complex_t));
            const int64_t n = sdhip_ndsp_psk_demod_work(h, (const float *)iblk.getSamples<complex_t>(), iblk.size, (float *)oblk.getSamples<complex_t>(), iblk.max_size);
            if (n < 0)
            {
                outputs[0].fifo->free(oblk);
                inputs[0].fifo->free(iblk);
                throw satdump_exception(std::string("sdhip: ") + sdhip_last_error());
            }
            oblk.size = (uint32_t)n;
            if (n > 0)
                outputs[0].fifo->wait_enqueue(oblk);
            else
                outputs[0].fifo->free(oblk);
            inputs[0].fifo->free(iblk);
            return false;
        }

    public:
        PSKDemodHipBlock() : Block("psk_demod_hip_cc", {{"in", satdump::ndsp::DSP_SAMPLE_TYPE_CF32}}, {{"out", satdump::ndsp::DSP_SAMPLE_TYPE_CF32}}) { sdhip_ndsp_psk_cfg_default(&cfg); }
        ~PSKDemodHipBlock() { drop(); }

        void init()
        {
            drop();
            h = sdhip_ndsp_psk_demod_create(&cfg);
            if (!h)
                throw satdump_exception(std::string("sdhip: ") + sdhip_last_error());
        }

        nlohmann::ordered_json get_cfg_list()
        { // psk_demod.h:108-158
            nlohmann::ordered_json v;
            satdump::ndsp::add_param_list(v, "constellation", "string", {"bpsk", "qpsk"});
            satdump::ndsp::add_param_simple(v, "samplerate", "float");
            satdump::ndsp::add_param_simple(v, "symbolrate", "float");
            satdump::ndsp::add_param_simple(v, "advanced", "bool");
            satdump::ndsp::add_param_simple(v, "device", "int");
            satdump::ndsp::add_param_simple(v, "exact", "bool");
            satdump::ndsp::add_param_simple(v, "pll_freq", "stat");
            if (advanced_mode)
                for (const char *k : {"rrc_gain", "rrc_alpha", "rrc_ntaps", "agc_rate", "agc_reference", "agc_gain", "agc_max_gain", "rec_omega", "rec_omegaGain", "rec_mu",
                                      "rec_muGain", "rec_omegaLimit", "rec_nfilt", "rec_ntaps", "pll_loop_bw", "pll_freq_limit"})
                    satdump::ndsp::add_param_simple(v, k, std::string(k).find("ntaps") != std::string::npos || std::string(k) == "rec_nfilt" ? "int" : "float");
            return v;
        }

        nlohmann::json get_cfg(std::string key)
        {
            if (key == "constellation")
                return constellation;
            else if (key == "samplerate")
                return cfg.samplerate;
            else if (key == "symbolrate")
                return cfg.symbolrate;
            else if (key == "advanced")
                return advanced_mode;
            else if (key == "device")
                return cfg.device;
            else if (key == "exact")
                return cfg.exact != 0;
            else if (key == "pll_freq")
            { // rad_to_hz(pll freq, symbolrate), psk_demod.h:170
                sdhip_demod_stats st{};
                if (h)
                    sdhip_ndsp_psk_demod_get_stats(h, &st);
                return (double)st.freq_hz;
            }
#define SDHIP_GET(k) \
    else if (key == #k) return cfg.k;
            SDHIP_GET(rrc_gain)
            SDHIP_GET(rrc_alpha)
            SDHIP_GET(rrc_ntaps)
            SDHIP_GET(agc_rate)
            SDHIP_GET(agc_reference)
            SDHIP_GET(agc_gain)
            SDHIP_GET(agc_max_gain)
            SDHIP_GET(rec_omega)
            SDHIP_GET(rec_omegaGain)
            SDHIP_GET(rec_mu)
            SDHIP_GET(rec_muGain)
            SDHIP_GET(rec_omegaLimit)
            SDHIP_GET(rec_nfilt)
            SDHIP_GET(rec_ntaps)
            SDHIP_GET(pll_loop_bw)
            SDHIP_GET(pll_freq_limit)
#undef SDHIP_GET
            else return nlohmann::ordered_json(); // psk_demod.h:196-199
        }

        cfg_res_t set_cfg(std::string key, nlohmann::json v)
        {
            if (key == "constellation" && (v == "bpsk" || v == "qpsk"))
            { // psk_demod.h:205-214
                constellation = v;
                cfg.constellation = v == "bpsk" ? SDHIP_BPSK : SDHIP_QPSK;
            }
            else if (key == "samplerate")
            {
                cfg.samplerate = v;
                cfg.rec_omega = 0; // setting either rate forces the clock recovery's omega back to samplerate / symbolrate (psk_demod.h:224)
            }
            else if (key == "symbolrate")
            {
                cfg.symbolrate = v;
                cfg.rec_omega = 0;
            }
            else if (key == "advanced")
            {
                advanced_mode = v;
                return RES_LISTUPD;
            }
            else if (key == "device")
                cfg.device = v;
            else if (key == "exact")
                cfg.exact = v.get<bool>() ? 1 : 0;
#define SDHIP_SET(k) \
    else if (key == #k) cfg.k = v;
            SDHIP_SET(rrc_gain)
            SDHIP_SET(rrc_alpha)
            SDHIP_SET(rrc_ntaps)
            SDHIP_SET(agc_rate)
            SDHIP_SET(agc_reference)
            SDHIP_SET(agc_gain)
            SDHIP_SET(agc_max_gain)
            SDHIP_SET(rec_omega)
            SDHIP_SET(rec_omegaGain)
            SDHIP_SET(rec_mu)
            SDHIP_SET(rec_muGain)
            SDHIP_SET(rec_omegaLimit)
            SDHIP_SET(rec_nfilt)
            SDHIP_SET(rec_ntaps)
            SDHIP_SET(pll_loop_bw)
            SDHIP_SET(pll_freq_limit)
#undef SDHIP_SET
            else return RES_ERR; // psk_demod.h:250-253
            // Applied at the next buffer, like MMClockRecoveryBlock / FIRBlock do (clock_recovery_mm.cpp:78-82) -- but as a re-creation of the WHOLE
            // engine: filter history, gain, clock and loop state start over and the filter's ntaps-sample latency is paid again, where the reference
            // re-initialises only the member block whose key was touched (and changes the Costas order live). A flowgraph configures its blocks
            // before start(); a key changed in mid-stream costs this block a re-acquisition the reference's would not have.
            needs_reinit = true;
            return RES_OK;
        }
    };
} // namespace sdhip_plugin
