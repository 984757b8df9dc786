// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI wrapper around the REFERENCE's own classes, compiled from the
// sources where they lie under /root/reference/src-core (see oracle/Makefile).
// Nothing in here is product code: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load the resulting oracle/_ref/libsdref.so.
//
// The module-level orchestration (which the reference keeps inside ImGui-laden
// module classes that do not compile standalone) is re-stated here, line for
// line, from:
//   pipeline/modules/demod/module_psk_demod.cpp:86-236      (psk_demod chain)
//   pipeline/modules/demod/module_demod_base.cpp:59-208     (resample decision, AGC)
//   pipeline/modules/ccsds/module_ccsds_conv_concat_decoder.cpp:140-200
//   plugins/noaa_metop_support/metop/module_metop_ahrpt_decoder.cpp:34-90
// DSP blocks are driven synchronously (producer swap -> block.work() -> ...),
// which is arithmetically identical to the reference's thread-per-block
// topology (dsp::stream is a strict hand-off, common/dsp/buffer.h:49-106).
//
// Built with -fno-access-control so Block::work() can be called directly.

#include <cstdint>
#include <cstring>
#include <cmath>
#include <memory>
#include <vector>
#include <new>
#include <atomic>
#include <chrono>
#include <thread>

#include "logger.h"
std::shared_ptr<slog::Logger> logger = std::make_shared<slog::Logger>();

#include "common/codings/viterbi/cc_decoder.h"
#include "common/codings/viterbi/cc_encoder.h"
#include "common/codings/viterbi/viterbi_1_2.h"
#include "common/codings/viterbi/viterbi_3_4.h"
#include "common/codings/viterbi/viterbi_punc.h"
#include "common/codings/rotation.h"
#include "common/codings/correlator.h"
#include "common/codings/viterbi/viterbi27.h"
#include "common/codings/differential/nrzm.h"
#include "common/codings/randomization.h"
#include "common/codings/differential/nrzm.h"
#include "common/codings/deframing/bpsk_ccsds_deframer.h"
#include "common/codings/reedsolomon/reedsolomon.h"
#include "common/codings/differential/qpsk_diff.h"
#include "fengyun3/diff.h" // plugins/fengyun3_support (the Makefile adds the include path and compiles fengyun3/diff.cpp)
#include "common/dsp/demod/constellation.h"

#include "common/dsp/block.h"
#include "common/dsp/utils/agc.h"
#include "common/codings/viterbi/viterbi27.h"
#include "common/dsp/utils/correct_iq.h"
#include "common/dsp/utils/freq_shift.h"
#include "common/dsp/filter/fir.h"
#include "common/dsp/filter/firdes.h"
#include "common/dsp/pll/costas_loop.h"
#include "common/dsp/pll/pll_carrier_tracking.h"
#include "common/dsp/utils/fast_trig.h"
#include "common/dsp/clock_recovery/clock_recovery_mm.h"
#include "common/dsp/clock_recovery/clock_recovery_gardner.h"
#include "common/dsp/resamp/rational_resampler.h"
#include "common/dsp/resamp/smart_resampler.h"
#include "common/dsp/window/window.h"
#include "common/dsp/demod/delay_one_imag.h"

#include "../include/sdhip.h" // plain-C config structs shared with the product ABI

namespace
{
    // Allocate an object on zero-filled storage. Viterbi1_2/3_4 read a few bytes
    // past ber_soft_buffer into ber_decoded_buffer (viterbi_1_2.cpp:68 with
    // cc_decoder.cpp:295-297: 2*(1024+6)+shift symbols from a 2048-byte member),
    // which is uninitialised in the reference. We pin that to zero.
    template <class T, class... A>
    T *zero_new(A... a)
    {
        void *p = calloc(1, sizeof(T));
        return new (p) T(a...);
    }
    template <class T>
    void zero_delete(T *p)
    {
        p->~T();
        free(p);
    }

    // sdref_pipeline_threaded: the decoder loops below run on their own thread and take their soft symbols from an array
    // the demodulator's consumer thread is still filling (the role of the module's output_fifo, pipeline_run.cpp:72-104).
    struct SoftFeed
    {
        std::atomic<int64_t> avail{0};
        std::atomic<bool> done{false};
        bool wait(int64_t need)
        {
            while (avail.load(std::memory_order_acquire) < need)
            {
                if (done.load(std::memory_order_acquire))
                    return avail.load(std::memory_order_acquire) >= need;
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            return true;
        }
    };
    thread_local SoftFeed *tl_feed = nullptr;
}

extern "C"
{
    // ------------------------------------------------------------------ unit level
    // Chained CCDecoder::work calls (cc_decoder.cpp:295-302). syms holds, per block,
    // 2*(frame+6) unsigned soft symbols (caller provides the tail). out: frame bits/block.
    void sdref_ccdecoder(int frame_bits, const uint8_t *syms, int nblocks, uint8_t *out)
    {
        viterbi::CCDecoder dec(frame_bits, 7, 2, {79, 109});
        const size_t stride = 2 * (size_t)(frame_bits + 6);
        std::vector<uint8_t> tmp(stride);
        for (int b = 0; b < nblocks; b++)
        {
            memcpy(tmp.data(), syms + b * stride, stride);
            dec.work(tmp.data(), out + (size_t)b * frame_bits);
        }
    }

    // CCEncoder::work (cc_encoder.cpp:92-104), one call over nbits, start state 0.
    void sdref_ccencode(const uint8_t *bits, int nbits, uint8_t *out)
    {
        viterbi::CCEncoder enc(nbits, 7, 2, {79, 109});
        enc.work((uint8_t *)bits, out);
    }

    // Viterbi27::work over nframes consecutive calls of one decoder (viterbi27.cpp:31-66), CCSDS polys
    void sdref_viterbi27(int frame_bits, int ber_test_size, const int8_t *soft, int nframes, uint8_t *out, float *ber)
    {
        viterbi::Viterbi27 *v = zero_new<viterbi::Viterbi27>(frame_bits, viterbi::CCSDS_R2_K7_POLYS, ber_test_size);
        for (int f = 0; f < nframes; f++)
        {
            v->work((int8_t *)soft + (size_t)f * 2 * frame_bits, out + (size_t)f * (frame_bits / 8));
            if (ber)
                ber[f] = v->ber();
        }
        zero_delete(v);
    }

    void sdref_derand(uint8_t *data, int len) { derand_ccsds(data, len); }

    // ReedSolomon::encode_interlaved / decode_interlaved (reedsolomon.cpp:53-143)
    void sdref_rs_encode(uint8_t *data, int nframes, int frame_stride, int dualbasis, int I, int rs239)
    {
        reedsolomon::ReedSolomon rs(rs239 ? reedsolomon::RS239 : reedsolomon::RS223);
        for (int f = 0; f < nframes; f++)
            rs.encode_interlaved(data + (size_t)f * frame_stride, dualbasis, I);
    }
    void sdref_rs_decode(uint8_t *data, int nframes, int frame_stride, int dualbasis, int I, int rs239, int fill_bytes, int *errors)
    {
        reedsolomon::ReedSolomon *rs = zero_new<reedsolomon::ReedSolomon>(rs239 ? reedsolomon::RS239 : reedsolomon::RS223, fill_bytes);
        for (int f = 0; f < nframes; f++)
            rs->decode_interlaved(data + (size_t)f * frame_stride, dualbasis, I, errors + (size_t)f * I);
        zero_delete(rs);
    }

    // BPSK_CCSDS_Deframer::work (bpsk_ccsds_deframer.cpp:24-107), fed in `chunk`-bit calls.
    int sdref_deframer(const uint8_t *bits, int64_t nbits, int chunk, int cadu_size, uint32_t asm_sync, int state_synced, uint8_t *out, int64_t out_cap_frames)
    {
        deframing::BPSK_CCSDS_Deframer def(cadu_size, asm_sync);
        def.STATE_SYNCED = state_synced;
        if (cadu_size % 8 != 0)
            def.CADU_PADDING = cadu_size % 8;
        const int cadu_bytes = (cadu_size + def.CADU_PADDING) / 8;
        std::vector<uint8_t> fb((size_t)(chunk / cadu_size + 2) * cadu_bytes + 64);
        int64_t nf = 0;
        for (int64_t p = 0; p < nbits; p += chunk)
        {
            int n = (int)std::min<int64_t>(chunk, nbits - p);
            int f = def.work((uint8_t *)bits + p, n, fb.data());
            for (int i = 0; i < f && nf < out_cap_frames; i++, nf++)
                memcpy(out + nf * cadu_bytes, fb.data() + (size_t)i * cadu_bytes, cadu_bytes);
        }
        return (int)nf;
    }

    // ------------------------------------------------------------------ FEC module level
    // In-memory restatement of CCSDSConvConcatDecoderModule::process()
    // (module_ccsds_conv_concat_decoder.cpp:140-200). Processes floor(n/buffer) blocks.
    // Optional taps: vit_bits (1 bit/byte, concatenated viterbi output after NRZ-M),
    // blk_ber/blk_state (per block, after work()), frm_err (rs_i ints per deframed frame,
    // before the usecheck filter), frm_pre (deframed frames before derand/RS).
    int64_t sdref_concat_decode(const sdhip_fec_cfg *c, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                                uint8_t *vit_bits, int64_t *vit_nbits, float *blk_ber, int *blk_state, int *frm_err, int64_t *n_deframed)
    {
        const int d_cadu_size = c->cadu_size;
        const int d_cadu_bytes = (int)ceil(d_cadu_size / 8.0);
        const int d_buffer_size = std::max<int>(d_cadu_size, 8192);
        const bool bpsk = c->constellation == SDHIP_BPSK || c->constellation == SDHIP_BPSK_90;
        const bool d_bpsk_90 = c->constellation == SDHIP_BPSK_90;
        const bool d_oqpsk_mode = c->constellation == SDHIP_OQPSK;
        std::vector<phase_t> d_phases;
        if (bpsk && !d_bpsk_90)
            d_phases = {PHASE_0};
        else if (bpsk && d_bpsk_90)
            d_phases = {PHASE_90};
        else
            d_phases = {PHASE_0, PHASE_90};

        viterbi::Viterbi1_2 *vit = zero_new<viterbi::Viterbi1_2>(c->viterbi_ber_thresold, c->viterbi_outsync_after, d_buffer_size, d_phases, d_oqpsk_mode);
        deframing::BPSK_CCSDS_Deframer deframer(d_cadu_size, c->asm_sync);
        if (d_cadu_size % 8 != 0)
            deframer.CADU_PADDING = d_cadu_size % 8;
        reedsolomon::ReedSolomon *rs = nullptr;
        if (c->rs_i != 0)
            rs = zero_new<reedsolomon::ReedSolomon>(c->rs_type == SDHIP_RS239 ? reedsolomon::RS239 : reedsolomon::RS223, c->rs_fill_bytes);

        std::vector<uint8_t> viterbi_out((size_t)d_buffer_size * 8, 0);
        std::vector<int8_t> soft_buffer(d_buffer_size);
        std::vector<uint8_t> frame_buffer((size_t)d_buffer_size * 8, 0);
        int errors[16] = {0};
        diff::NRZMDiff diff;

        int64_t nout = 0, nvb = 0, ndef = 0;
        const int64_t nblocks = n / d_buffer_size;
        for (int64_t b = 0; b < nblocks; b++)
        {
            if (tl_feed && !tl_feed->wait((b + 1) * (int64_t)d_buffer_size))
                break;
            memcpy(soft_buffer.data(), soft + b * d_buffer_size, d_buffer_size);
            if (d_bpsk_90 || c->iq_invert)
                rotate_soft(soft_buffer.data(), d_buffer_size, PHASE_0, true);
            int vitout = vit->work(soft_buffer.data(), d_buffer_size, viterbi_out.data());
            if (blk_ber)
                blk_ber[b] = vit->ber();
            if (blk_state)
                blk_state[b] = vit->getState();
            if (c->nrzm)
                diff.decode_bits(viterbi_out.data(), vitout);
            if (vit_bits)
                memcpy(vit_bits + nvb, viterbi_out.data(), vitout);
            nvb += vitout;
            int frames = deframer.work(viterbi_out.data(), vitout, frame_buffer.data());
            for (int i = 0; i < frames; i++)
            {
                uint8_t *cadu = &frame_buffer[(size_t)i * d_cadu_bytes];
                if (c->derandomize && !c->derand_after_rs)
                    derand_ccsds(&cadu[c->derand_start], d_cadu_bytes - c->derand_start);
                if (c->rs_i != 0)
                    rs->decode_interlaved(&cadu[4], c->rs_dualbasis, c->rs_i, errors);
                bool valid = true;
                for (int k = 0; k < c->rs_i; k++)
                    if (errors[k] == -1)
                        valid = false;
                if (frm_err)
                    for (int k = 0; k < c->rs_i; k++)
                        frm_err[ndef * c->rs_i + k] = errors[k];
                ndef++;
                if (c->derandomize && c->derand_after_rs)
                    derand_ccsds(&cadu[c->derand_start], d_cadu_bytes - c->derand_start);
                if (!c->rs_usecheck || valid)
                {
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * d_cadu_bytes, cadu, d_cadu_bytes);
                    nout++;
                }
            }
        }
        if (vit_nbits)
            *vit_nbits = nvb;
        if (n_deframed)
            *n_deframed = ndef;
        zero_delete(vit);
        if (rs)
            zero_delete(rs);
        return nout;
    }

    // CCSDSConvConcatDecoderModule::process() with conv_rate != "1/2" (module_ccsds_conv_concat_decoder.cpp:107-119,140-200):
    // the same loop around viterbi::Viterbi_Depunc. rate: 1 = 2/3, 2 = 3/4, 3 = 5/6, 4 = 7/8.
    // The decoder's heap buffers are uninitialised in the reference (new uint8_t[]); they are pinned to zero here like the
    // object storage itself (zero_new) -- with these rates the first decode already finds > buffer_size + 12 valid symbols in
    // the sliding buffer, so no uninitialised byte reaches the trellis anyway.
    int64_t sdref_concat_decode_punc(const sdhip_fec_cfg *c, int rate, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                                     uint8_t *vit_bits, int64_t *vit_nbits, float *blk_ber, int *blk_state, int *frm_err, int64_t *n_deframed)
    {
        const int d_cadu_size = c->cadu_size;
        const int d_cadu_bytes = (int)ceil(d_cadu_size / 8.0);
        const int d_buffer_size = std::max<int>(d_cadu_size, 8192);
        const bool bpsk = c->constellation == SDHIP_BPSK || c->constellation == SDHIP_BPSK_90;
        const bool d_bpsk_90 = c->constellation == SDHIP_BPSK_90;
        const bool d_oqpsk_mode = c->constellation == SDHIP_OQPSK;
        std::vector<phase_t> d_phases;
        if (bpsk && !d_bpsk_90)
            d_phases = {PHASE_0};
        else if (bpsk && d_bpsk_90)
            d_phases = {PHASE_90};
        else
            d_phases = {PHASE_0, PHASE_90};

        std::shared_ptr<viterbi::puncturing::GenericDepunc> dp;
        if (rate == 1)
            dp = std::make_shared<viterbi::puncturing::Depunc23>();
        else if (rate == 2)
            dp = std::make_shared<viterbi::puncturing::Depunc34>();
        else if (rate == 3)
            dp = std::make_shared<viterbi::puncturing::Depunc56>();
        else
            dp = std::make_shared<viterbi::puncturing::Depunc78>();
        viterbi::Viterbi_Depunc *vit = zero_new<viterbi::Viterbi_Depunc>(dp, c->viterbi_ber_thresold, c->viterbi_outsync_after, d_buffer_size, d_phases, d_oqpsk_mode);
        memset(vit->soft_buffer, 0, (size_t)d_buffer_size * 8);
        memset(vit->depunc_buffer, 0, (size_t)d_buffer_size * 8);
        memset(vit->output_buffer, 0, (size_t)d_buffer_size * 8);
        memset(vit->vit_buffer.buffer_ptr, 0, (size_t)d_buffer_size * 4);
        deframing::BPSK_CCSDS_Deframer deframer(d_cadu_size, c->asm_sync);
        if (d_cadu_size % 8 != 0)
            deframer.CADU_PADDING = d_cadu_size % 8;
        reedsolomon::ReedSolomon *rs = nullptr;
        if (c->rs_i != 0)
            rs = zero_new<reedsolomon::ReedSolomon>(c->rs_type == SDHIP_RS239 ? reedsolomon::RS239 : reedsolomon::RS223, c->rs_fill_bytes);

        std::vector<uint8_t> viterbi_out((size_t)d_buffer_size * 8, 0);
        std::vector<int8_t> soft_buffer(d_buffer_size);
        std::vector<uint8_t> frame_buffer((size_t)d_buffer_size * 8, 0);
        int errors[16] = {0};
        diff::NRZMDiff diff;

        int64_t nout = 0, nvb = 0, ndef = 0;
        const int64_t nblocks = n / d_buffer_size;
        for (int64_t b = 0; b < nblocks; b++)
        {
            memcpy(soft_buffer.data(), soft + b * d_buffer_size, d_buffer_size);
            if (d_bpsk_90 || c->iq_invert)
                rotate_soft(soft_buffer.data(), d_buffer_size, PHASE_0, true);
            int vitout = vit->work(soft_buffer.data(), d_buffer_size, viterbi_out.data());
            if (blk_ber)
                blk_ber[b] = vit->ber();
            if (blk_state)
                blk_state[b] = vit->getState();
            if (c->nrzm)
                diff.decode_bits(viterbi_out.data(), vitout);
            if (vit_bits)
                memcpy(vit_bits + nvb, viterbi_out.data(), vitout);
            nvb += vitout;
            int frames = deframer.work(viterbi_out.data(), vitout, frame_buffer.data());
            for (int i = 0; i < frames; i++)
            {
                uint8_t *cadu = &frame_buffer[(size_t)i * d_cadu_bytes];
                if (c->derandomize && !c->derand_after_rs)
                    derand_ccsds(&cadu[c->derand_start], d_cadu_bytes - c->derand_start);
                if (c->rs_i != 0)
                    rs->decode_interlaved(&cadu[4], c->rs_dualbasis, c->rs_i, errors);
                bool valid = true;
                for (int k = 0; k < c->rs_i; k++)
                    if (errors[k] == -1)
                        valid = false;
                if (frm_err)
                    for (int k = 0; k < c->rs_i; k++)
                        frm_err[ndef * c->rs_i + k] = errors[k];
                ndef++;
                if (c->derandomize && c->derand_after_rs)
                    derand_ccsds(&cadu[c->derand_start], d_cadu_bytes - c->derand_start);
                if (!c->rs_usecheck || valid)
                {
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * d_cadu_bytes, cadu, d_cadu_bytes);
                    nout++;
                }
            }
        }
        if (vit_nbits)
            *vit_nbits = nvb;
        if (n_deframed)
            *n_deframed = ndef;
        zero_delete(vit);
        if (rs)
            zero_delete(rs);
        return nout;
    }

    // In-memory restatement of CCSDSSimplePSKDecoderModule::process()
    // (pipeline/modules/ccsds/module_ccsds_simple_psk_decoder.cpp:104-296), soft symbols not "hard_symbols".
    // frm_err: rs_i ints per deframed frame (before the usecheck filter).
    int64_t sdref_simple_decode(const sdhip_fec_cfg *c, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames, int *frm_err,
                                int64_t *n_deframed)
    {
        const int d_cadu_size = c->cadu_size;
        const int d_cadu_bytes = (int)ceil(d_cadu_size / 8.0);
        const int d_buffer_size = d_cadu_size;
        const bool is_qpsk = c->constellation == SDHIP_QPSK;
        const bool m23 = c->oqpsk_method2 || c->oqpsk_method3;
        std::vector<uint8_t> bits_out((size_t)d_buffer_size * 2, 0); // indeterminate in the reference: taken as zero
        std::vector<int8_t> soft_buffer(d_buffer_size), soft_buffer2(d_buffer_size);
        std::vector<uint8_t> qpsk_diff_buffer((size_t)d_cadu_size * 2, 0);
        std::vector<uint8_t> frame_buffer((size_t)d_cadu_size * 2 + 4 * d_cadu_bytes, 0);
        deframing::BPSK_CCSDS_Deframer deframer(d_cadu_size, c->asm_sync), deframer_qpsk(d_cadu_size, c->asm_sync);
        if (d_cadu_size % 8 != 0)
        {
            deframer.CADU_PADDING = d_cadu_size % 8;
            deframer_qpsk.CADU_PADDING = d_cadu_size % 8;
        }
        reedsolomon::ReedSolomon *rs = nullptr;
        if (c->rs_i != 0)
            rs = zero_new<reedsolomon::ReedSolomon>(c->rs_type == SDHIP_RS239 ? reedsolomon::RS239 : reedsolomon::RS223, c->rs_fill_bytes);
        int errors[16] = {0};
        diff::NRZMDiff diff;
        diff::QPSKDiff *qpsk_diff = zero_new<diff::QPSKDiff>(); // buffer[] is uninitialised in the reference; its first two symbols are skipped anyway
        qpsk_diff->swap = c->qpsk_swap_diff;
        dsp::constellation_t qpsk_const(dsp::QPSK);
        int8_t last_oqpsk2 = 0, last_q_oqpsk = 0;
        int64_t nout = 0, ndef = 0;
        const int64_t nblocks = n / d_buffer_size;
        for (int64_t b = 0; b < nblocks; b++)
        {
            int frames = 0;
            memcpy(soft_buffer.data(), soft + b * d_buffer_size, d_buffer_size);
            if (!is_qpsk)
            {
                for (int i = 0; i < d_buffer_size; i++)
                    bits_out[i] = soft_buffer[i] > 0;
                if (c->nrzm)
                    diff.decode_bits(bits_out.data(), d_buffer_size);
            }
            else
            {
                if (c->oqpsk_delay)
                    for (int i = 0; i < d_buffer_size / 2; i++)
                    {
                        int8_t back = soft_buffer[i * 2 + 0];
                        soft_buffer[i * 2 + 0] = last_q_oqpsk;
                        last_q_oqpsk = back;
                    }
                if (c->qpsk_swap_iq)
                    rotate_soft(soft_buffer.data(), d_buffer_size, PHASE_0, true);
                auto demod_to_bits = [&](int8_t *sb) {
                    for (int i = 0; i < d_buffer_size / 2; i++)
                    {
                        uint8_t sym = qpsk_const.soft_demod(&sb[i * 2]);
                        bits_out[i * 2 + 0] = sym >> 1;
                        bits_out[i * 2 + 1] = sym & 1;
                    }
                };
                auto delay2 = [&]() {
                    memcpy(soft_buffer2.data(), soft_buffer.data(), d_buffer_size);
                    for (int i = 0; i < d_buffer_size / 2; i++)
                    {
                        int8_t back = soft_buffer2[i * 2 + 0];
                        soft_buffer2[i * 2 + 0] = last_oqpsk2;
                        last_oqpsk2 = back;
                    }
                };
                if (c->nrzm)
                {
                    for (int i = 0; i < d_buffer_size / 2; i++)
                        qpsk_diff_buffer[i] = qpsk_const.soft_demod(&soft_buffer[i * 2]);
                    qpsk_diff->work(qpsk_diff_buffer.data(), d_buffer_size / 2, bits_out.data());
                }
                else if (!m23)
                {
                    demod_to_bits(soft_buffer.data());
                    frames += deframer_qpsk.work(bits_out.data(), d_buffer_size, &frame_buffer[(size_t)frames * d_cadu_bytes]);
                    rotate_soft(soft_buffer.data(), d_buffer_size, PHASE_90, false);
                    demod_to_bits(soft_buffer.data());
                }
                else if (!c->oqpsk_method3)
                {
                    delay2();
                    demod_to_bits(soft_buffer2.data());
                    frames += deframer_qpsk.work(bits_out.data(), d_buffer_size, &frame_buffer[(size_t)frames * d_cadu_bytes]);
                    rotate_soft(soft_buffer.data(), d_buffer_size, PHASE_90, false);
                    demod_to_bits(soft_buffer.data());
                }
                else
                {
                    delay2();
                    rotate_soft(soft_buffer2.data(), d_buffer_size, PHASE_90, false);
                    demod_to_bits(soft_buffer2.data());
                    frames += deframer_qpsk.work(bits_out.data(), d_buffer_size, &frame_buffer[(size_t)frames * d_cadu_bytes]);
                    demod_to_bits(soft_buffer.data());
                }
            }
            frames += deframer.work(bits_out.data(), d_buffer_size, &frame_buffer[(size_t)frames * d_cadu_bytes]);
            for (int i = 0; i < frames; i++)
            {
                uint8_t *cadu = &frame_buffer[(size_t)i * d_cadu_bytes];
                if (c->derandomize && !c->derand_after_rs)
                    derand_ccsds(&cadu[c->derand_start], d_cadu_bytes - c->derand_start);
                if (c->rs_i != 0)
                    rs->decode_interlaved(&cadu[4], c->rs_dualbasis, c->rs_i, errors);
                bool valid = true;
                for (int k = 0; k < c->rs_i; k++)
                    if (errors[k] == -1)
                        valid = false;
                if (frm_err)
                    for (int k = 0; k < c->rs_i; k++)
                        frm_err[ndef * c->rs_i + k] = errors[k];
                ndef++;
                if (c->derandomize && c->derand_after_rs)
                    derand_ccsds(&cadu[c->derand_start], d_cadu_bytes - c->derand_start);
                if (!c->rs_usecheck || valid)
                {
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * d_cadu_bytes, cadu, d_cadu_bytes);
                    nout++;
                }
            }
        }
        if (n_deframed)
            *n_deframed = ndef;
        zero_delete(qpsk_diff);
        if (rs)
            zero_delete(rs);
        return nout;
    }

    // In-memory restatement of METEORLRPTDecoderModule::process(), the classic (non m2x_mode) branch
    // (plugins/meteor_support/meteor/module_meteor_lrpt_decoder.cpp:201-262), on the reference's own Correlator / Viterbi27 / NRZMDiff /
    // ReedSolomon. The input behaves like the module's file: read_data copies what is left (a short read keeps the buffer's old tail),
    // should_run() turns false once a read has hit the end. locks_out (may be NULL): `pos == 0` per iteration.
    int64_t sdref_lrpt_decode(int diff_decode, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames, int *locks_out, int64_t locks_cap,
                              int64_t *iterations_out)
    {
        const int FRAME_SIZE = 1024, ENCODED_FRAME_SIZE = 1024 * 8 * 2;
        std::vector<int8_t> store(ENCODED_FRAME_SIZE + 64, 0);
        int8_t *buffer = store.data() + 64;
        viterbi::Viterbi27 *viterbi = zero_new<viterbi::Viterbi27>(ENCODED_FRAME_SIZE / 2, viterbi::CCSDS_R2_K7_POLYS);
        int64_t rd = 0;
        bool eof = false;
        auto read_data = [&](uint8_t *dst, size_t len) {
            const size_t have = (size_t)std::min<int64_t>((int64_t)len, n - rd);
            memcpy(dst, soft + rd, have);
            rd += (int64_t)have;
            if (have < len)
                eof = true;
        };
        Correlator correlator(QPSK, diff_decode ? 0xfc4ef4fd0cc2df89 : 0xfca2b63db00d9794);
        reedsolomon::ReedSolomon rs(reedsolomon::RS223);
        uint8_t frameBuffer[1024 + 64] = {0};
        int errors[4] = {0, 0, 0, 0};
        phase_t phase = PHASE_0;
        bool swap = false;
        int cor = 0;
        diff::NRZMDiff diff;
        int64_t nout = 0, it = 0;
        while (!eof)
        {
            read_data((uint8_t *)buffer, ENCODED_FRAME_SIZE);
            int pos = correlator.correlate((int8_t *)buffer, phase, swap, cor, ENCODED_FRAME_SIZE);
            if (locks_out && it < locks_cap)
                locks_out[it] = pos == 0;
            it++;
            if (pos != 0 && pos < ENCODED_FRAME_SIZE)
            {
                std::memmove(buffer, &buffer[pos], ENCODED_FRAME_SIZE - pos);
                read_data((uint8_t *)&buffer[ENCODED_FRAME_SIZE - pos], pos);
            }
            rotate_soft(buffer, ENCODED_FRAME_SIZE, phase, swap);
            viterbi->work((int8_t *)buffer, frameBuffer);
            if (diff_decode)
                diff.decode(frameBuffer, FRAME_SIZE);
            derand_ccsds(&frameBuffer[4], FRAME_SIZE - 4);
            if (frameBuffer[9] == 0xFF)
                for (int i = 0; i < FRAME_SIZE; i++)
                    frameBuffer[i] ^= 0xFF;
            rs.decode_interlaved(&frameBuffer[4], false, 4, errors);
            if (errors[0] >= 0 && errors[1] >= 0 && errors[2] >= 0 && errors[3] >= 0)
            {
                if (nout < cadu_cap_frames)
                {
                    const uint8_t sync[4] = {0x1d, 0xcf, 0xfc, 0x1d};
                    memcpy(cadu_out + nout * FRAME_SIZE, sync, 4);
                    memcpy(cadu_out + nout * FRAME_SIZE + 4, &frameBuffer[4], FRAME_SIZE - 4);
                }
                nout++;
            }
        }
        if (iterations_out)
            *iterations_out = it;
        zero_delete(viterbi);
        return nout;
    }

    // In-memory restatement of MetOpAHRPTDecoderModule::process()
    // (plugins/noaa_metop_support/metop/module_metop_ahrpt_decoder.cpp:34-90).
    int64_t sdref_metop_decode(float ber_thr, int outsync_after, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                               uint8_t *vit_bits, int64_t *vit_nbits, float *blk_ber, int *blk_state, int *frm_err)
    {
        const int BUFFER_SIZE = 8192 * 2;
        viterbi::Viterbi3_4 *vit = zero_new<viterbi::Viterbi3_4>(ber_thr, outsync_after, BUFFER_SIZE, false);
        deframing::BPSK_CCSDS_Deframer deframer;
        deframer.STATE_SYNCED = 18;
        reedsolomon::ReedSolomon *rs = zero_new<reedsolomon::ReedSolomon>(reedsolomon::RS223, 0);
        std::vector<uint8_t> viterbi_out(BUFFER_SIZE * 2, 0);
        std::vector<int8_t> soft_buffer(BUFFER_SIZE);
        std::vector<uint8_t> frame_buffer(1024 * 10, 0);
        int errors[4] = {0, 0, 0, 0};
        int noSyncsRuns = 0;
        int64_t nout = 0, nvb = 0;
        const int64_t nblocks = n / BUFFER_SIZE;
        for (int64_t b = 0; b < nblocks; b++)
        {
            if (tl_feed && !tl_feed->wait((b + 1) * (int64_t)BUFFER_SIZE))
                break;
            memcpy(soft_buffer.data(), soft + b * BUFFER_SIZE, BUFFER_SIZE);
            int num_samp = vit->work(soft_buffer.data(), BUFFER_SIZE, viterbi_out.data());
            if (blk_ber)
                blk_ber[b] = vit->ber();
            if (blk_state)
                blk_state[b] = vit->getState();
            if (num_samp > 0)
            {
                if (vit_bits)
                    memcpy(vit_bits + nvb, viterbi_out.data(), num_samp);
                nvb += num_samp;
                int frames = deframer.work(viterbi_out.data(), num_samp, frame_buffer.data());
                if (deframer.getState() == deframer.STATE_NOSYNC)
                {
                    noSyncsRuns++;
                    if (noSyncsRuns >= 10)
                    {
                        vit->reset();
                        noSyncsRuns = 0;
                    }
                }
                else
                    noSyncsRuns = 0;
                for (int i = 0; i < frames; i++)
                {
                    uint8_t *cadu = &frame_buffer[i * 1024];
                    derand_ccsds(&cadu[4], 1024 - 4);
                    rs->decode_interlaved(&cadu[4], true, 4, errors);
                    if (frm_err)
                        for (int k = 0; k < 4; k++)
                            frm_err[nout * 4 + k] = errors[k];
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * 1024, cadu, 1024);
                    nout++;
                }
            }
        }
        if (vit_nbits)
            *vit_nbits = nvb;
        zero_delete(vit);
        zero_delete(rs);
        return nout;
    }

    // In-memory restatement of FengyunAHRPTDecoderModule::process() (plugins/fengyun3_support/fengyun3/module_fengyun_ahrpt_decoder.cpp:46-126) on the
    // reference's own Viterbi3_4 (fymode), FengyunDiff, BPSK_CCSDS_Deframer, derand_ccsds and ReedSolomon. With shift == 1 the module reads one pair past
    // its BUFFER_SIZE * 2 soft bytes (:64-67, i = BUFFER_SIZE - 1); the buffer here carries one more pair, kept at zero, which pins that read.
    // blk_ber / blk_state (may be NULL): 2 entries per read, Viterbi 1 then Viterbi 2; frm_err: 4 ints per frame.
    int64_t sdref_fy3_decode(float ber_thr, int outsync_after, int invert_second_viterbi, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                             float *blk_ber, int *blk_state, int *frm_err, int *shift_out, int *invert_branches_out)
    {
        const int BUFFER_SIZE = 8192;
        viterbi::Viterbi3_4 *viterbi1 = zero_new<viterbi::Viterbi3_4>(ber_thr, outsync_after, BUFFER_SIZE, true);
        viterbi::Viterbi3_4 *viterbi2 = zero_new<viterbi::Viterbi3_4>(ber_thr, outsync_after, BUFFER_SIZE, true);
        deframing::BPSK_CCSDS_Deframer deframer;
        deframer.STATE_SYNCING = 8;
        deframer.STATE_SYNCED = 16;
        fengyun3::FengyunDiff diff;
        reedsolomon::ReedSolomon *rs = zero_new<reedsolomon::ReedSolomon>(reedsolomon::RS223);
        std::vector<int8_t> soft_buffer(BUFFER_SIZE * 2 + 2, 0), i_soft_buffer(BUFFER_SIZE), q_soft_buffer(BUFFER_SIZE);
        std::vector<uint8_t> viterbi1_out(BUFFER_SIZE * 2, 0), viterbi2_out(BUFFER_SIZE * 2, 0), diff_out(BUFFER_SIZE * 20, 0), frame_buffer(1024 * 10, 0);
        int errors[4] = {0, 0, 0, 0};
        int shift = 0;
        bool iq_invert = true, invert_branches = false;
        int noSyncRuns = 0, viterbiNoSyncRun = 0;
        int64_t nout = 0;
        const int64_t nreads = n / (BUFFER_SIZE * 2);
        for (int64_t b = 0; b < nreads; b++)
        {
            memcpy(soft_buffer.data(), soft + b * BUFFER_SIZE * 2, BUFFER_SIZE * 2);
            rotate_soft(soft_buffer.data(), BUFFER_SIZE * 2, PHASE_0, iq_invert);
            for (int i = 0; i < BUFFER_SIZE; i++)
            {
                i_soft_buffer[i] = soft_buffer[(i + shift) * 2 + 0];
                q_soft_buffer[i] = invert_second_viterbi ? ~soft_buffer[(i + shift) * 2 + 1] : soft_buffer[(i + shift) * 2 + 1];
            }
            const int v1 = viterbi1->work(i_soft_buffer.data(), BUFFER_SIZE, viterbi1_out.data());
            const int v2 = viterbi2->work(q_soft_buffer.data(), BUFFER_SIZE, viterbi2_out.data());
            const int vout = std::min(v1, v2);
            if (blk_ber)
            {
                blk_ber[2 * b] = viterbi1->ber();
                blk_ber[2 * b + 1] = viterbi2->ber();
            }
            if (blk_state)
            {
                blk_state[2 * b] = viterbi1->getState();
                blk_state[2 * b + 1] = viterbi2->getState();
            }
            if (viterbi1->getState() == 0 || viterbi2->getState() == 0)
            {
                viterbiNoSyncRun++;
                if (viterbiNoSyncRun >= 10)
                    shift = shift == 0 ? 1 : 0;
            }
            diff.work2(invert_branches ? viterbi1_out.data() : viterbi2_out.data(), invert_branches ? viterbi2_out.data() : viterbi1_out.data(), vout, diff_out.data());
            if (v1 > 0 && v2 > 0)
            {
                int frames = deframer.work(diff_out.data(), vout * 2, frame_buffer.data());
                if (deframer.getState() == deframer.STATE_NOSYNC)
                {
                    noSyncRuns++;
                    if (noSyncRuns >= 10)
                    {
                        invert_branches = !invert_branches;
                        noSyncRuns = 0;
                    }
                }
                else
                    noSyncRuns = 0;
                for (int i = 0; i < frames; i++)
                {
                    uint8_t *cadu = &frame_buffer[i * 1024];
                    derand_ccsds(&cadu[4], 1024 - 4);
                    rs->decode_interlaved(&cadu[4], true, 4, errors);
                    if (frm_err)
                        for (int k = 0; k < 4; k++)
                            frm_err[nout * 4 + k] = errors[k];
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * 1024, cadu, 1024);
                    nout++;
                }
            }
        }
        if (shift_out)
            *shift_out = shift;
        if (invert_branches_out)
            *invert_branches_out = invert_branches ? 1 : 0;
        zero_delete(viterbi1);
        zero_delete(viterbi2);
        zero_delete(rs);
        return nout;
    }

    // In-memory restatement of FengyunMPTDecoderModule::process() (plugins/fengyun3_support/fengyun3/module_fengyun_mpt_decoder.cpp:43-134) on the reference's
    // own Viterbi1_2 (phases 0 / 90), FengyunDiff, BPSK_CCSDS_Deframer (default thresholds), derand_ccsds and ReedSolomon: the AHRPT module's loop with rate-1/2
    // rails, the second rail always complemented, a second I/Q exchange on each rail, and its watchdog condition as written (Viterbi 1's state twice, :78).
    int64_t sdref_fy3_mpt_decode(float ber_thr, int outsync_after, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames, float *blk_ber, int *blk_state,
                                 int *frm_err, int *shift_out, int *invert_branches_out)
    {
        const int BUFFER_SIZE = 8192;
        viterbi::Viterbi1_2 *viterbi1 = zero_new<viterbi::Viterbi1_2>(ber_thr, outsync_after, BUFFER_SIZE, std::vector<phase_t>{PHASE_0, PHASE_90});
        viterbi::Viterbi1_2 *viterbi2 = zero_new<viterbi::Viterbi1_2>(ber_thr, outsync_after, BUFFER_SIZE, std::vector<phase_t>{PHASE_0, PHASE_90});
        deframing::BPSK_CCSDS_Deframer deframer;
        fengyun3::FengyunDiff diff;
        reedsolomon::ReedSolomon *rs = zero_new<reedsolomon::ReedSolomon>(reedsolomon::RS223);
        std::vector<int8_t> soft_buffer(BUFFER_SIZE * 2 + 2, 0), i_soft_buffer(BUFFER_SIZE), q_soft_buffer(BUFFER_SIZE);
        std::vector<uint8_t> viterbi1_out(BUFFER_SIZE * 2, 0), viterbi2_out(BUFFER_SIZE * 2, 0), diff_out(BUFFER_SIZE * 20, 0), frame_buffer(1024 * 10, 0);
        int errors[4] = {0, 0, 0, 0};
        int shift = 0;
        bool iq_invert = true, invert_branches = false;
        int noSyncRuns = 0, viterbiNoSyncRun = 0;
        int64_t nout = 0;
        const int64_t nreads = n / (BUFFER_SIZE * 2);
        for (int64_t b = 0; b < nreads; b++)
        {
            memcpy(soft_buffer.data(), soft + b * BUFFER_SIZE * 2, BUFFER_SIZE * 2);
            rotate_soft(soft_buffer.data(), BUFFER_SIZE * 2, PHASE_0, iq_invert);
            for (int i = 0; i < BUFFER_SIZE; i++)
            {
                i_soft_buffer[i] = soft_buffer[(i + shift) * 2 + 0];
                q_soft_buffer[i] = ~soft_buffer[(i + shift) * 2 + 1];
            }
            rotate_soft(i_soft_buffer.data(), BUFFER_SIZE, PHASE_0, true);
            rotate_soft(q_soft_buffer.data(), BUFFER_SIZE, PHASE_0, true);
            const int v1 = viterbi1->work(i_soft_buffer.data(), BUFFER_SIZE, viterbi1_out.data());
            const int v2 = viterbi2->work(q_soft_buffer.data(), BUFFER_SIZE, viterbi2_out.data());
            const int vout = std::min(v1, v2);
            if (blk_ber)
            {
                blk_ber[2 * b] = viterbi1->ber();
                blk_ber[2 * b + 1] = viterbi2->ber();
            }
            if (blk_state)
            {
                blk_state[2 * b] = viterbi1->getState();
                blk_state[2 * b + 1] = viterbi2->getState();
            }
            if (viterbi1->getState() == 0 || viterbi1->getState() == 0)
            {
                viterbiNoSyncRun++;
                if (viterbiNoSyncRun >= 10)
                    shift = shift == 0 ? 1 : 0;
            }
            diff.work2(invert_branches ? viterbi1_out.data() : viterbi2_out.data(), invert_branches ? viterbi2_out.data() : viterbi1_out.data(), vout, diff_out.data());
            if (v1 > 0 && v2 > 0)
            {
                int frames = deframer.work(diff_out.data(), vout * 2, frame_buffer.data());
                if (deframer.getState() == deframer.STATE_NOSYNC)
                {
                    noSyncRuns++;
                    if (noSyncRuns >= 10)
                    {
                        invert_branches = !invert_branches;
                        noSyncRuns = 0;
                    }
                }
                else
                    noSyncRuns = 0;
                for (int i = 0; i < frames; i++)
                {
                    uint8_t *cadu = &frame_buffer[i * 1024];
                    derand_ccsds(&cadu[4], 1024 - 4);
                    rs->decode_interlaved(&cadu[4], true, 4, errors);
                    if (frm_err)
                        for (int k = 0; k < 4; k++)
                            frm_err[nout * 4 + k] = errors[k];
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * 1024, cadu, 1024);
                    nout++;
                }
            }
        }
        if (shift_out)
            *shift_out = shift;
        if (invert_branches_out)
            *invert_branches_out = invert_branches ? 1 : 0;
        zero_delete(viterbi1);
        zero_delete(viterbi2);
        zero_delete(rs);
        return nout;
    }

    // FengyunDiff::work2 alone (fengyun3/diff.cpp:49-78) over one call of len bit pairs, and its transmit-side inverse for the test streams: the
    // pairs (x, y) whose work2 output is the given dibit stream, found pair by pair from the decoder's own rule.
    void sdref_fy3_diff2(const uint8_t *in1, const uint8_t *in2, int len, uint8_t *out)
    {
        fengyun3::FengyunDiff diff;
        diff.work2((uint8_t *)in1, (uint8_t *)in2, len, out);
    }

    // ------------------------------------------------------------------ DSP unit level
    int sdref_rrc_taps(double gain, double fs, double symrate, double alpha, int ntaps, float *out)
    {
        std::vector<float> t = dsp::firdes::root_raised_cosine(gain, fs, symrate, alpha, ntaps);
        memcpy(out, t.data(), t.size() * sizeof(float));
        return (int)t.size();
    }
    // MM interpolator bank (clock_recovery_mm.cpp:18, polyphase_bank.cpp:6-39): out[nfilt][ntaps]
    int sdref_mm_bank(int nfilt, int ntaps, float *out)
    {
        dsp::PolyphaseBank pfb;
        pfb.init(dsp::windowed_sinc(nfilt * ntaps, dsp::hz_to_rad(0.5 / (double)nfilt, 1.0), dsp::window::nuttall, nfilt), nfilt);
        for (int i = 0; i < pfb.nfilt; i++)
            memcpy(out + (size_t)i * pfb.ntaps, pfb.taps[i], pfb.ntaps * sizeof(float));
        return pfb.ntaps;
    }
    // Rational resampler bank (rational_resampler.cpp:27-41): returns ntaps/phase, out[interp][ntaps]
    int sdref_resamp_bank(unsigned interp, unsigned decim, float *out, int cap, int *interp_red, int *decim_red)
    {
        dsp::RationalResamplerBlock<complex_t> r(nullptr, interp, decim);
        if (interp_red)
            *interp_red = r.d_interpolation;
        if (decim_red)
            *decim_red = r.d_decimation;
        if (r.pfb.nfilt * r.pfb.ntaps > cap)
            return -r.pfb.ntaps;
        for (int i = 0; i < r.pfb.nfilt; i++)
            memcpy(out + (size_t)i * r.pfb.ntaps, r.pfb.taps[i], r.pfb.ntaps * sizeof(float));
        return r.pfb.ntaps;
    }

    // Run ONE block type over a stream, fed in `chunk`-sample buffers. kind: see switch.
    // p[] carries the constructor arguments. Returns output sample count.
    int64_t sdref_block_run(int kind, const float *p, const float *in_c, int64_t n, int chunk, float *out_c, int64_t out_cap)
    {
        auto in = std::make_shared<dsp::stream<complex_t>>();
        std::shared_ptr<dsp::Block<complex_t, complex_t>> blk;
        std::shared_ptr<dsp::stream<complex_t>> outs;
        std::shared_ptr<dsp::AGCBlock<complex_t>> agc;
        std::shared_ptr<dsp::FIRBlock<complex_t>> fir;
        std::shared_ptr<dsp::CostasLoopBlock> pll;
        std::shared_ptr<dsp::MMClockRecoveryBlock<complex_t>> rec;
        std::shared_ptr<dsp::RationalResamplerBlock<complex_t>> rr;
        std::shared_ptr<dsp::CorrectIQBlock<complex_t>> dc;
        std::shared_ptr<dsp::DelayOneImagBlock> dly;
        std::shared_ptr<dsp::GardnerClockRecoveryBlock<complex_t>> gar;
        std::shared_ptr<dsp::PLLCarrierTrackingBlock> cpll;
        switch (kind)
        {
        case 0: // AGC(rate, ref, gain, max_gain)
            agc = std::make_shared<dsp::AGCBlock<complex_t>>(in, p[0], p[1], p[2], p[3]);
            outs = agc->output_stream;
            break;
        case 1: // FIR RRC(fs, symrate, alpha, ntaps)
            fir = std::make_shared<dsp::FIRBlock<complex_t>>(in, dsp::firdes::root_raised_cosine(1, p[0], p[1], p[2], (int)p[3]));
            outs = fir->output_stream;
            break;
        case 2: // Costas(loop_bw, order, freq_limit)
            pll = std::make_shared<dsp::CostasLoopBlock>(in, p[0], (unsigned)p[1], p[2]);
            outs = pll->output_stream;
            break;
        case 3: // MM(omega, omega_gain, mu, mu_gain, omega_limit)
            rec = std::make_shared<dsp::MMClockRecoveryBlock<complex_t>>(in, p[0], p[1], p[2], p[3], p[4]);
            outs = rec->output_stream;
            break;
        case 4: // Rational resampler(interp, decim)
            rr = std::make_shared<dsp::RationalResamplerBlock<complex_t>>(in, (unsigned)p[0], (unsigned)p[1]);
            outs = rr->output_stream;
            break;
        case 5: // CorrectIQ (DC block)
            dc = std::make_shared<dsp::CorrectIQBlock<complex_t>>(in);
            outs = dc->output_stream;
            break;
        case 6: // DelayOneImag
            dly = std::make_shared<dsp::DelayOneImagBlock>(in);
            outs = dly->output_stream;
            break;
        case 7: // Gardner(omega, omega_gain, mu, mu_gain, omega_limit)
            gar = std::make_shared<dsp::GardnerClockRecoveryBlock<complex_t>>(in, p[0], p[1], p[2], p[3], p[4]);
            memset(gar->buffer, 0, sizeof(complex_t) * 64); // volk_malloc'ed, read before written: taken as zero history
            gar->sample = gar->zc_sample = gar->last_sample = complex_t(0, 0); // uninitialised members in the reference
            outs = gar->output_stream;
            break;
        case 8: // PLLCarrierTracking(loop_bw, max, min)
            cpll = std::make_shared<dsp::PLLCarrierTrackingBlock>(in, p[0], p[1], p[2]);
            outs = cpll->output_stream;
            break;
        default:
            return -1;
        }
        int64_t no = 0;
        for (int64_t pos = 0; pos < n; pos += chunk)
        {
            int m = (int)std::min<int64_t>(chunk, n - pos);
            memcpy(in->writeBuf, in_c + 2 * pos, (size_t)m * sizeof(complex_t));
            in->swap(m);
            if (agc) agc->work();
            if (fir) fir->work();
            if (pll) pll->work();
            if (rec) rec->work();
            if (rr) rr->work();
            if (dc) dc->work();
            if (dly) dly->work();
            if (gar) gar->work();
            if (cpll) cpll->work();
            int k = outs->read();
            if (k > 0)
            {
                int64_t take = std::min<int64_t>(k, out_cap - no);
                memcpy(out_c + 2 * no, outs->readBuf, (size_t)take * sizeof(complex_t));
                no += take;
            }
            outs->flush();
        }
        return no;
    }

    // The psk_demod block chain, built exactly as BaseDemodModule::initb + PSKDemodModule::init build it.
    struct RefDemodChain
    {
        int d_buffer_size = 0;
        float final_sps = 0;
        bool is_bpsk = false, is_oqpsk = false, ok = true;
        std::shared_ptr<dsp::stream<complex_t>> in;
        std::shared_ptr<dsp::CorrectIQBlock<complex_t>> dc_blocker;
        std::shared_ptr<dsp::FreqShiftBlock> freq_shift;
        std::shared_ptr<dsp::SmartResamplerBlock<complex_t>> rresamp;
        std::shared_ptr<dsp::AGCBlock<complex_t>> agc;
        std::shared_ptr<dsp::FIRBlock<complex_t>> rrc;
        std::shared_ptr<dsp::PLLCarrierTrackingBlock> carrier_pll;
        std::shared_ptr<dsp::CorrectIQBlock<complex_t>> carrier_dc;
        std::shared_ptr<dsp::CostasLoopBlock> pll;
        std::shared_ptr<dsp::CorrectIQBlock<complex_t>> post_pll_dc;
        std::shared_ptr<dsp::DelayOneImagBlock> delay;
        std::shared_ptr<dsp::MMClockRecoveryBlock<complex_t>> rec;
        explicit RefDemodChain(const sdhip_demod_cfg *c)
        {
            // --- BaseDemodModule ctor + initb (module_demod_base.cpp:22-25, 59-89)
            long d_samplerate = (long)c->samplerate;
            int d_symbolrate = (int)c->symbolrate;
            d_buffer_size = c->buffer_size > 0 ? c->buffer_size : std::min<int>(dsp::STREAM_BUFFER_SIZE, std::max<int>(8192 + 1, d_samplerate / 200));
            float MIN_SPS = c->min_sps, MAX_SPS = c->max_sps;
            is_bpsk = c->constellation == SDHIP_BPSK;
            is_oqpsk = c->constellation == SDHIP_OQPSK;
            if (is_oqpsk)
            {
                MIN_SPS = 1.6;
                MAX_SPS = 2.4;
            }
            float input_sps = (float)d_samplerate / (float)d_symbolrate;
            bool resample = input_sps > MAX_SPS || input_sps < MIN_SPS;
            int range = pow(10, (std::to_string(int(d_symbolrate)).size() - 1));
            float final_samplerate = d_samplerate;
            if (c->custom_samplerate > 0) // d_parameters.count("custom_samplerate") > 0, module_demod_base.cpp:73-74
                final_samplerate = (long)c->custom_samplerate;
            else if (MAX_SPS == MIN_SPS)
                final_samplerate = d_symbolrate * MAX_SPS;
            else if (input_sps > MAX_SPS)
                final_samplerate = resample ? (round(d_symbolrate / range) * range) * MAX_SPS : d_samplerate;
            else if (input_sps < MIN_SPS)
                final_samplerate = resample ? d_symbolrate * MIN_SPS : d_samplerate;
            float decimation_factor = d_samplerate / final_samplerate;
            if (resample)
                d_buffer_size *= ceil(decimation_factor);
            if (d_buffer_size > 8192 * 20)
                d_buffer_size = 8192 * 20;
            final_sps = final_samplerate / (float)d_symbolrate;
            in = std::make_shared<dsp::stream<complex_t>>();
            std::shared_ptr<dsp::stream<complex_t>> cur = in;
            if (c->dc_block)
            {
                dc_blocker = std::make_shared<dsp::CorrectIQBlock<complex_t>>(cur);
                cur = dc_blocker->output_stream;
            }
            if (c->freq_shift != 0) // module_demod_base.cpp:122-123 (d_frequency_shift is a long)
            {
                freq_shift = std::make_shared<dsp::FreqShiftBlock>(cur, d_samplerate, (double)(long)c->freq_shift);
                cur = freq_shift->output_stream;
            }
            // SmartResamplerBlock(input, final_samplerate, d_samplerate) (module_demod_base.cpp:204): the reference's own class --
            // power-of-two pre-decimator (power_decim.cpp, its tap tables) + rational resampler as the ratio demands
            if (resample)
            {
                rresamp = std::make_shared<dsp::SmartResamplerBlock<complex_t>>(cur, final_samplerate, d_samplerate);
                cur = rresamp->output_stream;
            }
            agc = std::make_shared<dsp::AGCBlock<complex_t>>(cur, c->agc_rate, 1.0f, 1.0f, 65536);
            // --- PSKDemodModule::init (module_psk_demod.cpp:86-136)
            rrc = std::make_shared<dsp::FIRBlock<complex_t>>(agc->output_stream, dsp::firdes::root_raised_cosine(1, final_samplerate, d_symbolrate, c->rrc_alpha, c->rrc_taps));
            if (c->has_carrier) // module_psk_demod.cpp:93-113
            {
                if (!is_bpsk)
                {
                    ok = false;
                    return;
                }
                carrier_pll = std::make_shared<dsp::PLLCarrierTrackingBlock>(rrc->output_stream, c->carrier_pll_bw, c->carrier_pll_max_offset, -c->carrier_pll_max_offset);
                carrier_dc = std::make_shared<dsp::CorrectIQBlock<complex_t>>(carrier_pll->output_stream);
            }
            float costas_max_offset = c->has_carrier ? 0.2 : 1.0;
            if (c->costas_max_offset_hz > 0)
                costas_max_offset = dsp::hz_to_rad(c->costas_max_offset_hz, final_samplerate);
            unsigned order = is_bpsk ? 2 : (c->constellation == SDHIP_8PSK ? 8 : 4);
            pll = std::make_shared<dsp::CostasLoopBlock>(c->has_carrier ? carrier_dc->output_stream : rrc->output_stream, c->pll_bw, order, costas_max_offset);
            if (c->post_costas_dc) // module_psk_demod.cpp:127-134
                post_pll_dc = std::make_shared<dsp::CorrectIQBlock<complex_t>>(pll->output_stream);
            if (is_oqpsk)
                delay = std::make_shared<dsp::DelayOneImagBlock>(c->post_costas_dc ? post_pll_dc->output_stream : pll->output_stream);
            rec = std::make_shared<dsp::MMClockRecoveryBlock<complex_t>>(is_oqpsk ? delay->output_stream : (c->post_costas_dc ? post_pll_dc->output_stream : pll->output_stream),
                                                                               final_sps, c->clock_gain_omega, c->clock_mu, c->clock_gain_mu, c->clock_omega_relative_limit);
        }
    };

    // ------------------------------------------------------------------ demod module level
    // In-memory restatement of PSKDemodModule init()/process() for cf32 input
    // (module_psk_demod.cpp:86-236 + module_demod_base.cpp:59-208). Stream semantics:
    // all n samples are consumed in buffer_size chunks (last one short); the file
    // source's stale-tail quirk is NOT reproduced.
    // Outputs: soft (int8; 1/sym BPSK, 2/sym otherwise), optional syms (float pairs).
    int64_t sdref_psk_demod(const sdhip_demod_cfg *c, const float *iq, int64_t n, int8_t *soft, int64_t soft_cap, float *syms, int64_t syms_cap,
                            int *buffer_size_out, float *final_sps_out)
    {
        RefDemodChain ch(c);
        if (!ch.ok)
            return -2;
        const int d_buffer_size = ch.d_buffer_size;
        const bool is_bpsk = ch.is_bpsk;
        if (buffer_size_out)
            *buffer_size_out = d_buffer_size;
        if (final_sps_out)
            *final_sps_out = ch.final_sps;
        auto &in = ch.in;
        auto &dc_blocker = ch.dc_blocker;
        auto &freq_shift = ch.freq_shift;
        auto &rresamp = ch.rresamp;
        auto &agc = ch.agc;
        auto &rrc = ch.rrc;
        auto &carrier_pll = ch.carrier_pll;
        auto &carrier_dc = ch.carrier_dc;
        auto &pll = ch.pll;
        auto &post_pll_dc = ch.post_pll_dc;
        auto &delay = ch.delay;
        auto &rec = ch.rec;

        auto clampf = [](float x) -> int8_t { // module_demod_base.h:106-113
            if (x < -128.0)
                return -127;
            if (x > 127.0)
                return 127;
            return x;
        };

        int64_t nsoft = 0, nsym = 0;
        for (int64_t pos = 0; pos < n; pos += d_buffer_size)
        {
            int m = (int)std::min<int64_t>(d_buffer_size, n - pos);
            if (c->iq_swap)
            {
                for (int i = 0; i < m; i++)
                    in->writeBuf[i] = complex_t(iq[2 * (pos + i) + 1], iq[2 * (pos + i)]);
            }
            else
                memcpy(in->writeBuf, iq + 2 * pos, (size_t)m * sizeof(complex_t));
            in->swap(m);
            if (dc_blocker) dc_blocker->work();
            if (freq_shift) freq_shift->work();
            if (rresamp) rresamp->work();
            agc->work();
            rrc->work();
            if (carrier_pll) carrier_pll->work(), carrier_dc->work();
            pll->work();
            if (post_pll_dc) post_pll_dc->work();
            if (delay) delay->work();
            rec->work();
            int dat_size = rec->output_stream->read();
            if (dat_size > 0)
            {
                complex_t *rb = rec->output_stream->readBuf;
                for (int i = 0; i < dat_size; i++)
                {
                    if (syms && nsym < syms_cap)
                    {
                        syms[2 * nsym] = rb[i].real;
                        syms[2 * nsym + 1] = rb[i].imag;
                    }
                    nsym++;
                    if (is_bpsk)
                    {
                        if (nsoft < soft_cap)
                            soft[nsoft] = clampf(rb[i].real * 50);
                        nsoft++;
                    }
                    else
                    {
                        if (nsoft + 1 < soft_cap)
                        {
                            soft[nsoft] = clampf(rb[i].real * 100);
                            soft[nsoft + 1] = clampf(rb[i].imag * 100);
                        }
                        nsoft += 2;
                    }
                }
            }
            rec->output_stream->flush();
        }
        return nsoft;
    }

    // ------------------------------------------------------------------ the reference's own run-time topology (CPU baseline)
    // psk_demod and the decoder as the pipeline runs them (pipeline_run.cpp:72-104, block.h:49-53): every DSP block on its own
    // thread (Block::start), joined by dsp::stream hand-offs; the module thread drains the clock-recovery output, quantises
    // (module_psk_demod.cpp:199-220) and hands the int8 symbols to the decoder module on its own thread through a FIFO; the
    // calling thread plays the file source. decoder: 0 = ccsds_conv_concat_decoder, 1 = metop_ahrpt_decoder.
    // Returns CADUs decoded; *seconds = wall clock from the first buffer fed to the decoder's last frame; *threads = threads
    // that carried work (blocks + source + module + decoder). The tail of the stream that is still inside the block
    // hand-offs when the source ends is dropped by stop(), as in the reference: throughput, not parity, is what
    // this entry is for.
    // soft_keep (optional, 2 * n + 64 bytes): the soft symbols the module thread wrote, for the caller's parity checks.
    int64_t sdref_pipeline_threaded(const sdhip_demod_cfg *c, const sdhip_fec_cfg *f, int decoder, const float *iq, int64_t n, uint8_t *cadu_out,
                                    int64_t cadu_cap_frames, double *seconds, int *threads, int64_t *nsoft_out, int8_t *soft_keep)
    {
        RefDemodChain ch(c);
        if (!ch.ok)
            return -2;
        const bool is_bpsk = ch.is_bpsk;
        std::vector<int8_t> soft_own(soft_keep ? 0 : (size_t)(2 * n + 64));
        struct SoftView
        {
            int8_t *p;
            size_t n;
            int8_t &operator[](size_t i) { return p[i]; }
            int8_t *data() { return p; }
            size_t size() const { return n; }
        } soft{soft_keep ? soft_keep : soft_own.data(), (size_t)(2 * n + 64)};
        SoftFeed feed;
        auto clampf = [](float x) -> int8_t {
            if (x < -128.0)
                return -127;
            if (x > 127.0)
                return 127;
            return x;
        };
        int nthreads = 3; // source (caller) + module thread + decoder thread
        const auto t0 = std::chrono::steady_clock::now();
        if (ch.dc_blocker) ch.dc_blocker->start(), nthreads++;
        if (ch.freq_shift) ch.freq_shift->start(), nthreads++;
        if (ch.rresamp) ch.rresamp->start(), nthreads++;
        ch.agc->start(), nthreads++;
        ch.rrc->start(), nthreads++;
        if (ch.carrier_pll) ch.carrier_pll->start(), ch.carrier_dc->start(), nthreads += 2;
        ch.pll->start(), nthreads++;
        if (ch.post_pll_dc) ch.post_pll_dc->start(), nthreads++;
        if (ch.delay) ch.delay->start(), nthreads++;
        ch.rec->start(), nthreads++;
        std::atomic<bool> module_run{true};
        std::atomic<int64_t> last_progress_us{0};
        std::thread module_thread([&] {
            int64_t nsoft = 0;
            while (module_run.load())
            {
                int dat_size = ch.rec->output_stream->read();
                if (dat_size <= 0)
                    continue;
                complex_t *rb = ch.rec->output_stream->readBuf;
                if (is_bpsk)
                    for (int i = 0; i < dat_size; i++)
                        soft[nsoft++] = clampf(rb[i].real * 50);
                else
                    for (int i = 0; i < dat_size; i++)
                    {
                        soft[nsoft++] = clampf(rb[i].real * 100);
                        soft[nsoft++] = clampf(rb[i].imag * 100);
                    }
                ch.rec->output_stream->flush();
                feed.avail.store(nsoft, std::memory_order_release);
                last_progress_us.store(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
            }
        });
        int64_t ncadu = 0;
        std::chrono::steady_clock::time_point t_dec_end = t0;
        std::thread decoder_thread([&] {
            tl_feed = &feed;
            if (decoder == 1)
                ncadu = sdref_metop_decode(f->viterbi_ber_thresold, f->viterbi_outsync_after, soft.data(), (int64_t)soft.size(), cadu_out, cadu_cap_frames, nullptr, nullptr,
                                           nullptr, nullptr, nullptr);
            else
                ncadu = sdref_concat_decode(f, soft.data(), (int64_t)soft.size(), cadu_out, cadu_cap_frames, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            tl_feed = nullptr;
            t_dec_end = std::chrono::steady_clock::now();
        });
        // file source
        for (int64_t pos = 0; pos < n; pos += ch.d_buffer_size)
        {
            const int m = (int)std::min<int64_t>(ch.d_buffer_size, n - pos);
            if (c->iq_swap)
                for (int i = 0; i < m; i++)
                    ch.in->writeBuf[i] = complex_t(iq[2 * (pos + i) + 1], iq[2 * (pos + i)]);
            else
                memcpy(ch.in->writeBuf, iq + 2 * pos, (size_t)m * sizeof(complex_t));
            ch.in->swap(m);
        }
        // let the hand-offs drain: stop once the module thread has seen nothing new for 20 ms
        for (;;)
        {
            const int64_t now_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (now_us - last_progress_us.load() > 20000)
                break;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        module_run.store(false);
        if (ch.dc_blocker) ch.dc_blocker->stop();
        if (ch.freq_shift) ch.freq_shift->stop();
        if (ch.rresamp) ch.rresamp->stop();
        ch.agc->stop();
        ch.rrc->stop();
        if (ch.carrier_pll) ch.carrier_pll->stop(), ch.carrier_dc->stop();
        ch.pll->stop();
        if (ch.post_pll_dc) ch.post_pll_dc->stop();
        if (ch.delay) ch.delay->stop();
        ch.rec->stop();
        ch.rec->output_stream->stopReader();
        module_thread.join();
        feed.done.store(true, std::memory_order_release);
        decoder_thread.join();
        if (seconds)
            *seconds = std::chrono::duration<double>(t_dec_end - t0).count() - 0.020; // minus the idle time of the drain rule above
        if (threads)
            *threads = nthreads;
        if (nsoft_out)
            *nsoft_out = feed.avail.load();
        return ncadu;
    }
}
