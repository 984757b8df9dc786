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
#include <vector>

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
    // ---- the chain's member blocks as ndsp::Block's of their own: what the flowgraph registry offers next to the hier block (dsp_flowgraph_register.cpp:
    // "AGC/Agc CC", "Filter/RRC CC", "Clock Recovery/MM CC", "PLL/Costas"). Each takes the reference block's own keys (dsp/agc/agc.h:38-78,
    // dsp/filter/rrc.h:34-66, dsp/clock_recovery/clock_recovery_mm.h:70-130, dsp/pll/costas.h:55-90) plus "device" / "exact", and hands every DSPBuffer
    // to the C ABI's single-block handle (sdhip_ndsp_block_create); the block's state carries across buffers as in the reference.
    class SingleHipBlock : public satdump::ndsp::Block
    {
        struct Key
        {
            const char *name;
            int field; // index into the cfg (below)
            bool integer;
        };
        int kind;
        sdhip_ndsp_psk_cfg cfg;
        void *h = nullptr;
        bool needs_reinit = true;
        int order = 2;
        std::vector<Key> keys;
        double *fd(int f)
        {
            switch (f)
            {
            case 0:
                return &cfg.rrc_gain;
            case 1:
                return &cfg.samplerate;
            case 2:
                return &cfg.symbolrate;
            case 3:
                return &cfg.rrc_alpha;
            default:
                return nullptr;
            }
        }
        float *ff(int f)
        {
            switch (f)
            {
            case 10:
                return &cfg.agc_rate;
            case 11:
                return &cfg.agc_reference;
            case 12:
                return &cfg.agc_gain;
            case 13:
                return &cfg.agc_max_gain;
            case 20:
                return &cfg.rec_omega;
            case 21:
                return &cfg.rec_omegaGain;
            case 22:
                return &cfg.rec_mu;
            case 23:
                return &cfg.rec_muGain;
            case 24:
                return &cfg.rec_omegaLimit;
            case 30:
                return &cfg.pll_loop_bw;
            case 31:
                return &cfg.pll_freq_limit;
            default:
                return nullptr;
            }
        }
        int *fi(int f)
        {
            switch (f)
            {
            case 4:
                return &cfg.rrc_ntaps;
            case 25:
                return &cfg.rec_nfilt;
            case 26:
                return &cfg.rec_ntaps;
            default:
                return nullptr;
            }
        }
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
            {
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
        static const char *id_of(int kind)
        {
            static const char *ids[9] = {"psk_demod_hip_cc", "rrc_fir_hip_cc", "agc_hip_cc", "clock_recovery_mm_hip_cc", "costas_hip_cc", "clock_recovery_gardner_hip_cc",
                                         "agc_fast_hip_cc", "costas_fast_hip_cc", "fast_clock_recovery_mm_hip_cc"};
            return ids[kind];
        }
        explicit SingleHipBlock(int kind_) : Block(id_of(kind_), {{"in", satdump::ndsp::DSP_SAMPLE_TYPE_CF32}}, {{"out", satdump::ndsp::DSP_SAMPLE_TYPE_CF32}}), kind(kind_)
        {
            sdhip_ndsp_psk_cfg_default(&cfg);
            cfg.agc_reference = 1.0f; // the blocks' own defaults (agc.h:15, clock_recovery_mm.h:17, costas.h:14-16): the hier block is what sets 0.6
            cfg.rec_omega = 2.0f;
            cfg.samplerate = 6e6;
            cfg.symbolrate = 2e6;
            if (kind == SDHIP_NDSP_RRC_FIR)
                keys = {{"gain", 0, false}, {"samplerate", 1, false}, {"symbolrate", 2, false}, {"alpha", 3, false}, {"ntaps", 4, true}};
            else if (kind == SDHIP_NDSP_AGC || kind == SDHIP_NDSP_AGC_FAST) // (agc_fast.h:45-84: the same keys)
                keys = {{"rate", 10, false}, {"reference", 11, false}, {"gain", 12, false}, {"max_gain", 13, false}};
            else if (kind == SDHIP_NDSP_MM_FAST) // (clock_recovery_mm_fast.h:59-116: the M&M block's keys without the bank's shape)
                keys = {{"omega", 20, false}, {"omegaGain", 21, false}, {"mu", 22, false}, {"muGain", 23, false}, {"omegaLimit", 24, false}};
            else if (kind == SDHIP_NDSP_MM || kind == SDHIP_NDSP_GARDNER) // (clock_recovery_gardner.h:57-130: the same keys)
                keys = {{"omega", 20, false}, {"omegaGain", 21, false}, {"mu", 22, false}, {"muGain", 23, false}, {"omegaLimit", 24, false}, {"nfilt", 25, true}, {"ntaps", 26, true}};
            else
                keys = {{"loop_bw", 30, false}, {"freq_limit", 31, false}};
        }
        ~SingleHipBlock() { drop(); }
        void init()
        {
            drop();
            h = sdhip_ndsp_block_create(kind, &cfg);
            if (!h)
                throw satdump_exception(std::string("sdhip: ") + sdhip_last_error());
        }
        nlohmann::ordered_json get_cfg_list()
        {
            nlohmann::ordered_json v;
            if (kind == SDHIP_NDSP_COSTAS || kind == SDHIP_NDSP_COSTAS_FAST)
                satdump::ndsp::add_param_simple(v, "order", "int");
            for (auto &k : keys)
                satdump::ndsp::add_param_simple(v, k.name, k.integer ? "int" : "float");
            satdump::ndsp::add_param_simple(v, "device", "int");
            satdump::ndsp::add_param_simple(v, "exact", "bool");
            return v;
        }
        nlohmann::json get_cfg(std::string key)
        {
            if (key == "device")
                return cfg.device;
            if (key == "exact")
                return cfg.exact != 0;
            if ((kind == SDHIP_NDSP_COSTAS || kind == SDHIP_NDSP_COSTAS_FAST) && key == "order")
                return order;
            for (auto &k : keys)
                if (key == k.name)
                {
                    if (double *d = fd(k.field))
                        return *d;
                    if (float *f = ff(k.field))
                        return *f;
                    return *fi(k.field);
                }
            throw satdump_exception(key); // the member blocks throw on unknown keys (agc.h:62)
        }
        cfg_res_t set_cfg(std::string key, nlohmann::json v)
        {
            if (key == "device")
                cfg.device = v;
            else if (key == "exact")
                cfg.exact = v.get<bool>() ? 1 : 0;
            else if ((kind == SDHIP_NDSP_COSTAS || kind == SDHIP_NDSP_COSTAS_FAST) && key == "order")
            {
                order = v;
                if (order != 2 && order != 4 && order != 8)
                    throw satdump_exception("costas order must be 2, 4 or 8");
                cfg.constellation = order == 2 ? SDHIP_BPSK : (order == 4 ? SDHIP_QPSK : SDHIP_8PSK);
            }
            else
            {
                bool found = false;
                for (auto &k : keys)
                    if (key == k.name)
                    {
                        if (double *d = fd(k.field))
                            *d = v;
                        else if (float *f = ff(k.field))
                            *f = v;
                        else
                            *fi(k.field) = v;
                        found = true;
                    }
                if (!found)
                    throw satdump_exception(key);
            }
            needs_reinit = true; // (one stage: re-creating the handle re-initialises exactly the block whose key was touched, as the reference does)
            return RES_OK;
        }
    };
} // namespace sdhip_plugin
