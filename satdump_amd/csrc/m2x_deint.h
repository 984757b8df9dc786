// m2x_deint.h -- the front of meteor_lrpt_decoder's `m2x_mode` + `interleaved` branch (plugins/meteor_support/meteor/module_meteor_lrpt_decoder.cpp:103-146): what its two
// meteor::DeinterleaverReader objects (plugins/meteor_support/meteor/deint.cpp) make of the .soft stream, on the device. Included by fec_engine.hip only.
//
// The branch as it stands in the reference tree never decodes (its sample reader reports an error after the first 8192 bytes: tests/test_lrpt_m2x_reference_cpu.py);
// this is the loop with that one token put right -- the reader returns what it read -- which is what the classes do when they are fed
// (oracle/ref_wrap_lrpt_m2x.cpp, reader_returns = 1) and what the device path is held to.
//
// DeinterleaverReader::read_samples(.., dst, 8192), per call c (deint.cpp:174-243): take num_samples = 8192 + 8 x (sync markers in between) samples from the stream
// into dst -- the stream being the .soft file for reader 1 and the file turned a quarter turn, 8192 bytes at a time, for reader 2 (DintSampleReader, :59-98) --, find
// the 8-sample marker that precedes every 72 data samples by an autocorrelation over the hard bits (:22-98: its offset against where the de-interleaver expects the
// next marker, and the constellation's rotation from the marker's averaged bits), take `offset` more samples or give that many back, turn the samples by `rotation`,
// and push them through the convolutional de-interleaver (:100-133: 36 branches, delays of 36 x 2048 samples apart, markers skipped). Put together:
//   * a call consumes the stream window [p_c, p_c + num_samples_c + offset_c) and the next one starts at p_c + num_samples_c + offset_c (the `from_prev` cache and the
//     extra read are two ways of saying that); num_samples_c follows from c alone (the de-interleaver's branch counter advances 8192 per call whatever the offsets);
//   * the de-interleaver's ring is written once per lap at every address, so its output sample t (t = 8192 c + j) IS data sample n = t - (35 - t % 36) x 73 728 of the
//     data stream (zero while n < 0), and data sample n = 8192 c' + i is byte offset_c' + i + 8 x (markers skipped up to it) of call c''s rotated window.
// So with the per-call descriptors (p_c, offset_c, rotation_c) known, every output sample is a GATHER from the raw stream: k_m2x_gather. The descriptors need the
// autocorrelation of every call's window, and a window's position depends on the offsets before it: the calls of a batch are placed on the assumption offset = 0 (the
// steady state: markers where they are expected), k_m2x_autocorr evaluates deint.cpp's autocorrelate() for all of them -- a thread per call, the function statement for
// statement --, and the host walks the results: the first call whose offset is not 0 moves everything behind it, which is then evaluated again.
// Memory the class reads without having written it is modelled as zero (the 80 bytes in front of dst, read when a negative offset steps back: the module leaves them
// as `new` returned them) or as the stream's own next byte (the partner of the last sample of an odd-sized window under a 90 / 270 degree rotation: stale in the class).
#pragma once
#include "common.h"
#include <vector>

namespace sdhip
{
    constexpr int M2X_STRIDE = 80, M2X_INTERSAMPS = 72, M2X_BRANCHES = 36, M2X_DELAY = 2048 * 36, M2X_LEN = 8192;
    constexpr int M2X_HARD_MAX = 1200; // bytes of hard bits of one window (num_samples <= 8192 + 8 * 115)

    struct M2xRead
    {
        long long p; // stream position of dst[0] of the call
        int off;     // `offset` the call's autocorrelation gave (signed)
        int rot;     // `rotation` after it
        int ns;      // num_samples
        int pad;
    };

    // stream value at absolute position x: reader 1 = the file, reader 2 = rotate_soft(chunk, 8192, PHASE_90) of the file's 8192-byte chunks (rotation.cpp:9-43: -128 -> -127
    // over the chunk, then (a, b) -> (b, -a) pair by pair); beyond the end of what has been fetched the FIFO holds zeros (buffer1.resize zero-fills, :66-71)
    __device__ __forceinline__ int m2x_raw(const signed char *raw, long long raw_base, long long raw_end, long long x)
    {
        return (x >= raw_base && x < raw_end) ? (int)raw[x - raw_base] : 0;
    }
    __device__ __forceinline__ int m2x_clamp(int v) { return v == -128 ? -127 : v; }
    __device__ __forceinline__ int m2x_stream(const signed char *raw, long long raw_base, long long raw_end, int second, long long x)
    {
        if (!second)
            return m2x_raw(raw, raw_base, raw_end, x);
        if ((x & 1) == 0)
            return m2x_clamp(m2x_raw(raw, raw_base, raw_end, x + 1));
        return -m2x_clamp(m2x_raw(raw, raw_base, raw_end, x - 1));
    }

    // deint.cpp:22-98, one thread per call. res[2 c] = best_idx, res[2 c + 1] = rotation. hard_scratch: M2X_HARD_MAX bytes per call.
    __global__ __launch_bounds__(64) void k_m2x_autocorr(const signed char *raw, long long raw_base, long long raw_end, int second, const long long *pos, const int *nsamp, int ncalls,
                                                          unsigned char *hard_scratch, int *res)
    {
        const int c = (int)(blockIdx.x * 64 + threadIdx.x);
        if (c >= ncalls)
            return;
        const long long p = pos[c];
        const int ns = nsamp[c];
        unsigned char *hard = hard_scratch + (size_t)c * M2X_HARD_MAX;
        // soft_to_hard(hard, dst, num_samples & ~7), :153-172: first sample at bit 7
        const int nb = (ns & ~7) / 8;
        for (int b = 0; b < nb; b++)
        {
            unsigned v = 0;
            for (int i = 7; i >= 0; i--)
                v |= (m2x_stream(raw, raw_base, raw_end, second, p + 8 * b + (7 - i)) < 0 ? 1u : 0u) << i;
            hard[b] = (unsigned char)v;
        }
        const int period = M2X_STRIDE / 8;
        int len = ns / 8;
        int ones_count[8 * (M2X_STRIDE / 8)], average_bit[8 * (M2X_STRIDE / 8) + 8];
        for (int i = 0; i < 8 * period; i++)
            ones_count[i] = 0;
        for (int i = 0; i < 8 * period + 8; i++)
            average_bit[i] = 0;
        len -= len % period;
        for (int i = 0; i < period; i++)
        {
            int j = len - period + i - 1;
            unsigned char tmp = hard[j];
            for (j -= period; j >= 0; j -= period)
            {
                const unsigned char x = hard[j] ^ tmp;
                tmp = hard[j];
                hard[j] = x;
                for (int k = 0; k < 8; k++)
                    average_bit[8 * i + 7 - k] += (tmp & (1 << k)) ? 1 : -1;
            }
        }
        unsigned char window = 0;
        for (int i = 0; i < 8 * (len - period); i++)
        {
            const unsigned char h = hard[i >> 3];
            window = (unsigned char)((window >> 1) | ((h << (i % 8)) & 0x80));
            ones_count[i % (8 * period)] += __popc((unsigned)window);
        }
        int best_idx = 0, best_corr = ones_count[0] - len / 64;
        for (int i = 1; i < 8 * period; i++)
            if (ones_count[i] < best_corr)
            {
                best_corr = ones_count[i];
                best_idx = i;
            }
        unsigned tmpb = 0;
        for (int i = 7; i >= 0; i--)
            tmpb |= (average_bit[best_idx + i] > 0 ? 1u << i : 0u);
        const unsigned sync[4] = {0x27, 0x4E, 0xD8, 0xB1};
        int rotation = 0, bc = __popc(tmpb ^ sync[0]);
        for (int i = 1; i < 4; i++)
        {
            const int corr = __popc(tmpb ^ sync[i]);
            if (bc > corr)
            {
                bc = corr;
                rotation = i;
            }
        }
        res[2 * c] = best_idx;
        res[2 * c + 1] = rotation;
    }

    // byte q of call r's window after rotate_soft(dst, size, rotation, false), size = ns + off (rotation.cpp:9-63)
    __device__ __forceinline__ int m2x_rotated(const signed char *raw, long long raw_base, long long raw_end, int second, const M2xRead &r, int q)
    {
        if (q < 0)
            return 0; // the bytes in front of dst
        const int size = r.ns + r.off;
        const long long P = r.p + q;
        auto s = [&](long long x, int qq) { // window byte qq as the rotation loop finds it: clamped inside the window, as it lies behind it
            const int v = m2x_stream(raw, raw_base, raw_end, second, x);
            return qq < size ? m2x_clamp(v) : v;
        };
        switch (r.rot)
        {
        case 0:
            return s(P, q);
        case 1:
            return (q & 1) == 0 ? s(P + 1, q + 1) : -s(P - 1, q - 1);
        case 2:
            return -s(P, q);
        default:
            return (q & 1) == 0 ? (int)(signed char)(-s(P + 1, q + 1)) : s(P - 1, q - 1);
        }
    }

    // outputs of calls [c0, c0 + ncalls): out[(c - c0) * 8192 + j]. reads[k] describes call rbase + k; every call an output can come from must be in there
    // (those not older than 35 x 73 728 data samples) -- older ones read as zero, like the ring's initial content
    __global__ __launch_bounds__(256) void k_m2x_gather(const signed char *raw, long long raw_base, long long raw_end, int second, const M2xRead *reads, long long rbase, int nreads,
                                                         long long c0, int ncalls, signed char *out)
    {
        const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
        if (g >= (long long)ncalls * M2X_LEN)
            return;
        const long long t = c0 * M2X_LEN + g;
        const long long n = t - (long long)(M2X_BRANCHES - 1 - (int)(t % M2X_BRANCHES)) * M2X_DELAY;
        int v = 0;
        if (n >= 0)
        {
            const long long cin = n / M2X_LEN;
            const int i = (int)(n % M2X_LEN);
            if (cin >= rbase && cin < rbase + nreads)
            {
                const M2xRead r = reads[cin - rbase];
                const int cur0 = (int)((cin * M2X_LEN) % M2X_INTERSAMPS);
                const int first = (M2X_INTERSAMPS - cur0) % M2X_INTERSAMPS; // the first data sample of the call in front of which a marker is skipped
                const int skips = i >= first ? (i - first) / M2X_INTERSAMPS + 1 : 0;
                v = m2x_rotated(raw, raw_base, raw_end, second, r, r.off + i + 8 * skips);
            }
        }
        out[g] = (signed char)v;
    }

    // one DeinterleaverReader: the calls made so far
    struct M2xBranch
    {
        int second = 0;
        long long calls = 0;  // read_samples calls completed
        long long p_next = 0; // stream position the next call's window starts at
        int rotation = 0;
        std::vector<M2xRead> hist; // descriptors of calls [hist_base, calls)
        long long hist_base = 0;
        static int num_samples(long long c)
        { // deinterleave_num_samples(8192), deint.cpp:135-146, with _cur_branch = (8192 c) % 72
            const int cur = (int)((c * M2X_LEN) % M2X_INTERSAMPS);
            const int num_syncs = (cur ? 0 : 1) + (M2X_LEN - (M2X_INTERSAMPS - cur) + M2X_INTERSAMPS - 1) / M2X_INTERSAMPS;
            return M2X_LEN + 8 * num_syncs;
        }
        static int expected_sync_offset(long long c)
        { // deinterleave_expected_sync_offset, :148-151
            const int cur = (int)((c * M2X_LEN) % M2X_INTERSAMPS);
            return cur ? M2X_INTERSAMPS - cur : 0;
        }
        // `offset` of a call from its autocorrelation's best_idx, :214-220
        static int offset_of(long long c, int best_idx)
        {
            int off = (best_idx - expected_sync_offset(c) + M2X_INTERSAMPS + 1) % M2X_STRIDE;
            return off > M2X_STRIDE / 2 ? off - M2X_STRIDE : off;
        }
    };
} // namespace sdhip
