// oracle/ref_wrap_dvbs2_demap.cpp -- TEST INFRASTRUCTURE ONLY: the reference's DVB-S2 PLFRAME -> soft bits stage, dvbs2::S2BBToSoft
// (plugins/dvb_support/dvbs2/dvbs2_bb_to_soft.{h,cpp}: PLS decode, PL descrambling, LUT soft demapping, the de-interleaver), driven frame by
// frame through its own dsp::stream input / output, and the objects DVBS2DemodModule::init builds around it (module_dvbs2_demod.cpp:85-126:
// get_dvbs2_cfg, constellation_t + make_lut(256), S2Deinterleaver). Compiled with -fno-access-control: work() is private, and the block's
// scratch buffer (`new int8_t[64800]`, never initialised by the reference) is zeroed so that the positions its pilots branch never
// writes are defined.
#include "dvbs2/dvbs2_bb_to_soft.h"
#include "dvbs2/dvbs2_pl_sync.h"
#include "dvbs2/dvbs2_pll.h"
#include "codings/dvb-s2/modcod_to_cfg.h"
#include <cstring>

extern "C"
{
    // -> {frame_slot_count, constellation (dvbs2_constellation_t), coderate, bits per symbol}
    int sdref_s2_cfg(int modcod, int shortframes, int pilots, int *out4)
    {
        try
        {
            auto cfg = dvbs2::get_dvbs2_cfg(modcod, shortframes, pilots);
            dsp::constellation_t c(cfg.constel_obj_type, cfg.g1, cfg.g2);
            out4[0] = cfg.frame_slot_count;
            out4[1] = (int)cfg.constellation;
            out4[2] = (int)cfg.coderate;
            out4[3] = c.getBitsCnt();
            return 0;
        }
        catch (std::exception &)
        {
            return -1;
        }
    }

    // the soft-demapper table constellation_t::make_lut(resolution) builds: out[x][y][bit]
    int sdref_s2_lut(int modcod, int shortframes, int resolution, int8_t *out)
    {
        auto cfg = dvbs2::get_dvbs2_cfg(modcod, shortframes, false);
        dsp::constellation_t c(cfg.constel_obj_type, cfg.g1, cfg.g2);
        c.make_lut(resolution);
        const int bits = c.getBitsCnt();
        for (int x = 0; x < resolution; x++)
            for (int y = 0; y < resolution; y++)
                for (int b = 0; b < bits; b++)
                    out[((size_t)x * resolution + y) * bits + b] = c.lut[x][y].bits[b];
        return bits;
    }

    // the host libm's atan2f, element-wise (what complex_t::arg() calls): the yardstick of the device restatement
    int sdref_atan2f(const float *y, const float *x, int n, float *out)
    {
        for (int i = 0; i < n; i++)
            out[i] = atan2f(y[i], x[i]);
        return 0;
    }

    // the demapper table's phase errors: constellation_t::make_lut(resolution)'s [x][y].phase_error
    int sdref_s2_lut_phase(int modcod, int shortframes, int resolution, float *out)
    {
        auto cfg = dvbs2::get_dvbs2_cfg(modcod, shortframes, false);
        dsp::constellation_t c(cfg.constel_obj_type, cfg.g1, cfg.g2);
        c.make_lut(resolution);
        for (int x = 0; x < resolution; x++)
            for (int y = 0; y < resolution; y++)
                out[(size_t)x * resolution + y] = c.lut[x][y].phase_error;
        return 0;
    }

    // dvbs2::S2PLLBlock (dvbs2_pll.{h,cpp}) as DVBS2DemodModule::init sets it up (module_dvbs2_demod.cpp:111-118), one work() per frame through its
    // own streams. frames: nframes x frame_stride complex floats in, the same layout out (only the symbols the block writes are copied:
    // the return value per frame). state_out = {phase, freq} behind the last frame.
    int sdref_s2_pll_from(int modcod, int shortframes, int pilots, float loop_bw, const float *frames, int frame_stride, int nframes, float *out, float *state_io);
    int sdref_s2_pll(int modcod, int shortframes, int pilots, float loop_bw, const float *frames, int frame_stride, int nframes, float *out, float *state_out)
    {
        state_out[0] = state_out[1] = 0.0f;
        return sdref_s2_pll_from(modcod, shortframes, pilots, loop_bw, frames, frame_stride, nframes, out, state_out);
    }
    // the same with the loop state set beforehand (state_io = {phase, freq} in and out): what a study of frame-parallel schedules needs
    int sdref_s2_pll_from(int modcod, int shortframes, int pilots, float loop_bw, const float *frames, int frame_stride, int nframes, float *out, float *state_io)
    {
        auto cfg = dvbs2::get_dvbs2_cfg(modcod, shortframes, pilots);
        auto in = std::make_shared<dsp::stream<complex_t>>();
        dvbs2::S2PLLBlock blk(in, loop_bw);
        blk.pilots = pilots;
        blk.constellation = std::make_shared<dsp::constellation_t>(cfg.constel_obj_type, cfg.g1, cfg.g2);
        blk.constellation->make_lut(256);
        blk.frame_slot_count = cfg.frame_slot_count;
        blk.pls_code = modcod << 2 | shortframes << 1 | pilots;
        blk.update();
        blk.phase = state_io[0];
        blk.freq = state_io[1];
        float *state_out = state_io;
        const int walked = (cfg.frame_slot_count + 1) * 90 + blk.pilot_cnt * 36;
        for (int f = 0; f < nframes; f++)
        {
            memcpy(in->writeBuf, frames + (size_t)f * frame_stride * 2, (size_t)frame_stride * sizeof(complex_t));
            in->swap(frame_stride);
            blk.work();
            const int got = blk.output_stream->read();
            if (got != frame_stride)
                return -2;
            memcpy(out + (size_t)f * frame_stride * 2, blk.output_stream->readBuf, (size_t)walked * sizeof(complex_t));
            blk.output_stream->flush();
        }
        state_out[0] = blk.phase;
        state_out[1] = blk.freq;
        return walked;
    }

    // dvbs2::S2PLSyncBlock (dvbs2_pl_sync.{h,cpp}): the symbols go into the block's ring buffer, work2() is called for as long as the ring holds
    // two frames' worth (it reads one frame, then up to one more frame's worth to re-align). frames_out: nframes x raw_frame_size complex
    // floats; consumed_out[k] = symbols frame k took out of the ring (raw_frame_size + its best_pos). Returns the frames written.
    int sdref_s2_pl_sync(int slot_number, int pilots, float thresold, const float *syms, long long nsyms, float *frames_out, int max_frames, int *consumed_out, int *raw_frame_size)
    {
        auto in = std::make_shared<dsp::stream<complex_t>>();
        dvbs2::S2PLSyncBlock blk(in, slot_number, pilots);
        blk.thresold = thresold;
        const int raw = blk.raw_frame_size;
        *raw_frame_size = raw;
        const complex_t *s = (const complex_t *)syms;
        long long fed = 0;
        int nf = 0;
        for (;;)
        {
            const int w = (int)std::min<long long>(nsyms - fed, std::min<long long>(blk.ring_buffer.getWritable(), 1 << 20));
            if (w > 0)
            {
                blk.ring_buffer.write((complex_t *)s + fed, w);
                fed += w;
            }
            if (blk.ring_buffer.getReadable() < 2 * raw)
            {
                if (fed >= nsyms)
                    break;
                continue;
            }
            if (nf >= max_frames)
                break;
            const int before = blk.ring_buffer.getReadable();
            blk.work2();
            const int got = blk.output_stream->read();
            if (got != raw)
                return -2;
            memcpy(frames_out + (size_t)nf * raw * 2, blk.output_stream->readBuf, (size_t)raw * sizeof(complex_t));
            blk.output_stream->flush();
            consumed_out[nf] = before - blk.ring_buffer.getReadable();
            nf++;
        }
        return nf;
    }

    // frames: nframes x frame_stride complex floats (what S2PLLBlock hands over: 90 header symbols, then the slots). out: nframes x
    // frame_slot_count * 90 * bits soft bits; pls[f] = {best_header} as S2BBToSoft::work decodes it.
    int sdref_s2_bb_to_soft(int modcod, int shortframes, int pilots, const float *frames, int frame_stride, int nframes, int8_t *out, int *pls)
    {
        auto cfg = dvbs2::get_dvbs2_cfg(modcod, shortframes, pilots);
        auto in = std::make_shared<dsp::stream<complex_t>>();
        dvbs2::S2BBToSoft blk(in);
        memset(blk.soft_slots_buffer, 0, 64800);
        blk.pilots = pilots;
        blk.constellation = std::make_shared<dsp::constellation_t>(cfg.constel_obj_type, cfg.g1, cfg.g2);
        blk.constellation->make_lut(256);
        blk.frame_slot_count = cfg.frame_slot_count;
        blk.deinterleaver = std::make_shared<dvbs2::S2Deinterleaver>(cfg.constellation, cfg.framesize, cfg.coderate);
        const int nout = cfg.frame_slot_count * 90 * blk.constellation->getBitsCnt();
        for (int f = 0; f < nframes; f++)
        {
            memcpy(in->writeBuf, frames + (size_t)f * frame_stride * 2, (size_t)frame_stride * sizeof(complex_t));
            in->swap(frame_stride);
            blk.work();
            const int got = blk.output_stream->read();
            if (got != nout)
                return -2;
            memcpy(out + (size_t)f * nout, blk.output_stream->readBuf, nout);
            blk.output_stream->flush();
            if (pls)
                pls[f] = (blk.detect_modcod << 2) | (blk.detect_shortframes ? 2 : 0) | (blk.detect_pilots ? 1 : 0);
        }
        return nout;
    }
}
