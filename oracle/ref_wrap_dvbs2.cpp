// oracle/ref_wrap_dvbs2.cpp -- TEST INFRASTRUCTURE ONLY: C entry points around the reference's own DVB-S2 FEC classes, compiled
// where they lie (oracle/Makefile `ref`): BBFrameLDPC (plugins/dvb_support/codings/dvb-s2/bbframe_ldpc.{h,cpp}: layered
// offset-min-sum decoder, int8, ldpc/layered_decoder.hh + algorithms.hh) and BBFrameBCH (bbframe_bch.{h,cpp}).
// Built twice: libsdref_dvbs2.so (no -msse4.1: SIMD<int8_t, 1>, one frame per decoder call) and libsdref_dvbs2_sse.so
// (-msse4.1, what plugins/dvb_support/CMakeLists.txt:25-38 builds on x86: SIMD<int8_t, 16>, 16 frames per call with ONE early exit).
#include "codings/dvb-s2/bbframe_ldpc.h"
#include "codings/dvb-s2/bbframe_bch.h"
#include "codings/dvb-s2/bbframe_descramble.h"
#include "codings/dvb-s2/s2_deinterleaver.h"
#include <cstring>
#include <vector>

#include "common/codings/dvb-s2/bbframe_ts_parser.h"

extern "C"
{
    // dvbs2::BBFrameTSParser fed the way S2TStoTCPModule::process feeds it (module_s2_ts_extractor.cpp:77-105): one frame per work() call, the 188 000-byte output
    // array of the module. Returns the packets written to out (cap packets).
    long long sdref_s2_ts_extract(int bbframe_bits, const unsigned char *frames, int nframes, unsigned char *out, long long cap_packets)
    {
        dvbs2::BBFrameTSParser ts_extractor(bbframe_bits);
        static unsigned char ts_frames[188 * 1000];
        std::vector<unsigned char> bb((size_t)bbframe_bits); // (the module's bb_buffer is bbframe_size BYTES-as-bits long: bbframe_size / 8 of it are read per frame)
        long long n = 0;
        for (int k = 0; k < nframes; k++)
        {
            memcpy(bb.data(), frames + (size_t)k * (bbframe_bits / 8), (size_t)(bbframe_bits / 8));
            const int cnt = ts_extractor.work(bb.data(), 1, ts_frames, 188 * 1000);
            for (int i = 0; i < cnt; i++)
            {
                if (n < cap_packets)
                    memcpy(out + n * 188, &ts_frames[i * 188], 188);
                n++;
            }
        }
        return n;
    }

    int sdref_ldpc_batch() { return dvbs2::simd_type::SIZE; }

    // framesize: 0 normal / 1 short; rate: dvbs2_code_rate_t. -> {N, K}
    int sdref_ldpc_dims(int framesize, int rate, int *n, int *k)
    {
        dvbs2::BBFrameLDPC dec((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        *n = dec.get_instance()->code_len();
        *k = dec.get_instance()->data_len();
        return 0;
    }

    // frames: nbatches * SIZE frames of N int8 soft bits, decoded in place batch by batch exactly like DVBS2DemodModule::process_s2
    // (module_dvbs2_demod.cpp:246-257). trials_out[b] = BBFrameLDPC::decode's return value for batch b (-1: not converged).
    int sdref_ldpc_decode(int framesize, int rate, int8_t *frames, int nbatches, int max_trials, int *trials_out)
    {
        dvbs2::BBFrameLDPC dec((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        const int N = dec.get_instance()->code_len();
        for (int b = 0; b < nbatches; b++)
            trials_out[b] = dec.decode(frames + (size_t)b * N * dvbs2::simd_type::SIZE, max_trials);
        return 0;
    }

    // BBFrameLDPC::encode (bbframe_ldpc.cpp:126-145): frame = N/8 bytes, the first K/8 hold the data; parity bytes are written behind
    int sdref_ldpc_encode(int framesize, int rate, uint8_t *frames, int nframes)
    {
        dvbs2::BBFrameLDPC enc((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        const int N = enc.get_instance()->code_len();
        for (int f = 0; f < nframes; f++)
            enc.encode(frames + (size_t)f * (N / 8));
        return 0;
    }

    int sdref_s2_deinterleave(int constellation, int framesize, int rate, int8_t *in, int8_t *out, int nframes)
    {
        dvbs2::S2Deinterleaver d((dvbs2::dvbs2_constellation_t)constellation, (dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        const int n = framesize == 0 ? 64800 : 16200;
        for (int f = 0; f < nframes; f++)
            d.deinterleave(in + (size_t)f * n, out + (size_t)f * n);
        return 0;
    }
    int sdref_bb_descramble(int framesize, int rate, uint8_t *frames, int nframes, int stride)
    {
        dvbs2::BBFrameDescrambler d((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        for (int f = 0; f < nframes; f++)
            d.work(frames + (size_t)f * stride);
        return 0;
    }
    int sdref_bch_dims(int framesize, int rate, int *kbch)
    {
        dvbs2::BBFrameBCH b((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        *kbch = b.dataSize();
        return 0;
    }
    // BBFrameBCH::encode / decode on packed frames of `stride` bytes each (bbframe_bch.cpp); corrections_out[f] = decode's return value
    int sdref_bch_encode(int framesize, int rate, uint8_t *frames, int nframes, int stride)
    {
        dvbs2::BBFrameBCH b((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        for (int f = 0; f < nframes; f++)
            b.encode(frames + (size_t)f * stride);
        return 0;
    }
    int sdref_bch_decode(int framesize, int rate, uint8_t *frames, int nframes, int stride, int *corrections_out)
    {
        dvbs2::BBFrameBCH b((dvbs2::dvbs2_framesize_t)framesize, (dvbs2::dvbs2_code_rate_t)rate);
        for (int f = 0; f < nframes; f++)
            corrections_out[f] = b.decode(frames + (size_t)f * stride);
        return 0;
    }
}
