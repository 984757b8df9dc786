// oracle/ref_wrap_lrpt_m2x.cpp -- TEST INFRASTRUCTURE (never shipped, never linked by the product): METEORLRPTDecoderModule::process(), its m2x_mode branch
// (plugins/meteor_support/meteor/module_meteor_lrpt_decoder.cpp:103-199), in memory, on the reference's own classes compiled in place: meteor::DeinterleaverReader
// (plugins/meteor_support/meteor/deint.cpp), viterbi::Viterbi1_2, deframing::BPSK_CCSDS_Deframer, diff::NRZMDiff, derand_ccsds, reedsolomon::ReedSolomon.
//
// What this wrapper is there to establish (round 6): in the reference tree this repo is built against, the INTERLEAVED branch cannot decode. The module hands its
// DintSampleReader an input_function that returns `false` (:125-129); DintSampleReader::read_more() takes `!input_function(..)` as "error" (:68), so after the
// FIRST 8192-byte read iserror is set for good, read1 / read2 return 0 from then on (:79-82, 90-93), and every DeinterleaverReader::read_samples call leaves at
// deint.cpp:190-194 before it has rotated or de-interleaved anything. The two Viterbis then run on buffers nobody writes; in file mode nothing is read any more, so
// should_run() (= !eof) never turns false. `reader_returns` = 0 restates that loop as it is (bounded by max_iterations); = 1 is the loop with the one token changed
// (input_function returns true), i.e. what the classes do when they are fed -- the behaviour a device path of this branch can be held to.
#include "common/codings/deframing/bpsk_ccsds_deframer.h"
#include "common/codings/differential/nrzm.h"
#include "common/codings/randomization.h"
#include "common/codings/reedsolomon/reedsolomon.h"
#include "common/codings/rotation.h"
#include "common/codings/viterbi/viterbi_1_2.h"
#include "meteor/deint.h"
#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

namespace
{
    // DintSampleReader (module_meteor_lrpt_decoder.cpp:59-98): two FIFOs filled 8192 bytes at a time, the second a quarter turn on
    class SampleReader
    {
        bool iserror = false;
        std::vector<int8_t> buffer1, buffer2;
        void read_more()
        {
            buffer1.resize(buffer1.size() + 8192);
            iserror = iserror || !input_function(&buffer1[buffer1.size() - 8192], 8192);
            buffer2.resize(buffer2.size() + 8192);
            memcpy(&buffer2[buffer2.size() - 8192], &buffer1[buffer1.size() - 8192], 8192);
            rotate_soft(&buffer2[buffer2.size() - 8192], 8192, PHASE_90, false);
        }

    public:
        std::function<int(int8_t *, size_t)> input_function;
        int read1(int8_t *buf, size_t len)
        {
            while (buffer1.size() < len && !iserror)
                read_more();
            if (iserror)
                return 0;
            memcpy(buf, buffer1.data(), len);
            buffer1.erase(buffer1.begin(), buffer1.begin() + len);
            return (int)len;
        }
        int read2(int8_t *buf, size_t len)
        {
            while (buffer2.size() < len && !iserror)
                read_more();
            if (iserror)
                return 0;
            memcpy(buf, buffer2.data(), len);
            buffer2.erase(buffer2.begin(), buffer2.begin() + len);
            return (int)len;
        }
    };
}

#include <new>
#include <cstdlib>

extern "C"
{
    // soft: the .soft file; cadu_out: cap frames of 1024 bytes. taps (may be NULL, taps_cap entries each): per iteration the Viterbi taken (1 / 2; 1 when not
    // interleaved), its state and ber() after the read. consumed_out: bytes the module's read_data has taken from the file when the loop ends.
    // Returns the CADUs written. (Both buffers are zero-filled here where the module leaves them as `new` returns them.)
    int64_t sdref_lrpt_m2x_decode(int diff_decode, int interleaved, int reader_returns, float ber_thr, int outsync_after, const int8_t *soft, int64_t n, int64_t max_iterations,
                                  uint8_t *cadu_out, int64_t cadu_cap_frames, int *tap_which, int *tap_state, float *tap_ber, int64_t taps_cap, int64_t *iterations_out,
                                  int64_t *consumed_out)
    {
        const int BUFFER_SIZE = 8192, ENCODED_FRAME_SIZE = 1024 * 8 * 2;
        std::vector<int8_t> store1(ENCODED_FRAME_SIZE + INTER_MARKER_STRIDE, 0), store2(ENCODED_FRAME_SIZE + INTER_MARKER_STRIDE, 0);
        int8_t *buffer = store1.data() + INTER_MARKER_STRIDE, *buffer2 = store2.data() + INTER_MARKER_STRIDE;
        std::vector<phase_t> phases = {PHASE_0, PHASE_90};
        // (on zero-filled storage: the class reads a few bytes of its ber_decoded_buffer member before it has written them, viterbi_1_2.cpp:68 -- pinned to zero
        // as in ref_wrap.cpp's zero_new; on the module's heap they are whatever was there)
        auto zero_vit = [&]() {
            void *p = calloc(1, sizeof(viterbi::Viterbi1_2));
            return std::shared_ptr<viterbi::Viterbi1_2>(new (p) viterbi::Viterbi1_2(ber_thr, outsync_after, BUFFER_SIZE, phases, true), [](viterbi::Viterbi1_2 *v) {
                v->~Viterbi1_2();
                free(v);
            });
        };
        auto viterbin = zero_vit(), viterbin2 = zero_vit();
        auto deframer = std::make_shared<deframing::BPSK_CCSDS_Deframer>(8192);
        std::shared_ptr<meteor::DeinterleaverReader> deint1, deint2;
        if (interleaved)
        {
            deint1 = std::make_shared<meteor::DeinterleaverReader>();
            deint2 = std::make_shared<meteor::DeinterleaverReader>();
        }
        std::vector<uint8_t> viterbi_out(BUFFER_SIZE * 2, 0), viterbi_out2(BUFFER_SIZE * 2, 0), frame_buffer(BUFFER_SIZE * 2, 0);
        reedsolomon::ReedSolomon reed_solomon(reedsolomon::RS223);
        diff::NRZMDiff diff;
        int errors[4] = {0, 0, 0, 0};
        int64_t rd = 0;
        bool eof = false;
        auto read_data = [&](uint8_t *dst, size_t len) { // the module's file: a short read leaves the tail of dst as it was and sets eof
            const size_t have = (size_t)std::min<int64_t>((int64_t)len, n - rd);
            memcpy(dst, soft + rd, have);
            rd += (int64_t)have;
            if (have < len)
                eof = true;
        };
        SampleReader file_reader;
        file_reader.input_function = [&](int8_t *buf, size_t len) -> int
        {
            read_data((uint8_t *)buf, len);
            return reader_returns ? true : false; // :128 has `return false;`
        };
        int64_t nout = 0, it = 0;
        while (!eof && it < max_iterations)
        {
            if (interleaved)
            {
                deint1->read_samples([&file_reader](int8_t *buf, size_t len) -> int { return (bool)file_reader.read1(buf, len); }, buffer, 8192);
                deint2->read_samples([&file_reader](int8_t *buf, size_t len) -> int { return (bool)file_reader.read2(buf, len); }, buffer2, 8192);
            }
            else
                read_data((uint8_t *)buffer, BUFFER_SIZE);
            int vitout = 0, vitout1 = 0, vitout2 = 0, which = 1, state = 0;
            float ber = 10;
            if (interleaved)
            {
                vitout1 = viterbin->work((int8_t *)buffer, BUFFER_SIZE, viterbi_out.data());
                vitout2 = viterbin2->work((int8_t *)buffer2, BUFFER_SIZE, viterbi_out2.data());
                if (viterbin2->getState() > viterbin->getState())
                {
                    vitout = vitout2;
                    ber = viterbin2->ber();
                    state = viterbin2->getState();
                    which = 2;
                    memcpy(viterbi_out.data(), viterbi_out2.data(), vitout2);
                }
                else
                {
                    vitout = vitout1;
                    ber = viterbin->ber();
                    state = viterbin->getState();
                }
            }
            else
            {
                vitout = viterbin->work((int8_t *)buffer, BUFFER_SIZE, viterbi_out.data());
                ber = viterbin->ber();
                state = viterbin->getState();
            }
            if (it < taps_cap)
            {
                if (tap_which)
                    tap_which[it] = which;
                if (tap_state)
                    tap_state[it] = state;
                if (tap_ber)
                    tap_ber[it] = ber;
            }
            it++;
            if (diff_decode)
                diff.decode_bits(viterbi_out.data(), vitout);
            const int frames = deframer->work(viterbi_out.data(), vitout, frame_buffer.data());
            for (int i = 0; i < frames; i++)
            {
                uint8_t *cadu = &frame_buffer[i * 1024];
                derand_ccsds(&cadu[4], 1020);
                reed_solomon.decode_interlaved(&cadu[4], false, 4, errors);
                if (errors[0] >= 0 && errors[1] >= 0 && errors[2] >= 0 && errors[3] >= 0)
                {
                    if (nout < cadu_cap_frames)
                        memcpy(cadu_out + nout * 1024, cadu, 1024);
                    nout++;
                }
            }
        }
        if (iterations_out)
            *iterations_out = it;
        if (consumed_out)
            *consumed_out = rd;
        return nout;
    }

    // One DeinterleaverReader fed from a sample array (its reader returns the length asked for while samples last, 0 afterwards): `reads` calls of
    // read_samples(.., dst, 8192), the 8192 outputs of each appended to out. pre_rotate = 1: the samples pass rotate_soft(PHASE_90) in 8192-byte pieces first
    // (DintSampleReader's second FIFO). rot_out / off_out (may be NULL): the class's `rotation` and cached `offset` after each call. Returns the calls that
    // completed (read_samples returned 0).
    int64_t sdref_m2x_deint(const int8_t *soft, int64_t n, int pre_rotate, int64_t reads, int8_t *out, int *rot_out, int *off_out)
    {
        std::vector<int8_t> src(soft, soft + n);
        if (pre_rotate)
            for (int64_t a = 0; a + 8192 <= n; a += 8192)
                rotate_soft(src.data() + a, 8192, PHASE_90, false);
        std::vector<int8_t> store(1024 * 8 * 2 + INTER_MARKER_STRIDE, 0);
        int8_t *dst = store.data() + INTER_MARKER_STRIDE;
        meteor::DeinterleaverReader *d = new meteor::DeinterleaverReader();
        int64_t rd = 0, done = 0;
        auto reader = [&](int8_t *buf, size_t len) -> int
        {
            if (rd + (int64_t)len > n)
                return 0;
            memcpy(buf, src.data() + rd, len);
            rd += (int64_t)len;
            return (int)len;
        };
        for (int64_t c = 0; c < reads; c++)
        {
            if (d->read_samples(reader, dst, 8192) != 0)
                break;
            memcpy(out + c * 8192, dst, 8192);
            if (rot_out)
                rot_out[c] = (int)d->rotation;
            if (off_out)
                off_out[c] = d->offset;
            done++;
        }
        delete d;
        return done;
    }
}
