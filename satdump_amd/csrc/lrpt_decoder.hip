// lrpt_decoder.hip -- METEORLRPTDecoderModule::process(), the classic (non "m2x_mode") branch
// (plugins/meteor_support/meteor/module_meteor_lrpt_decoder.cpp:201-262): a soft-symbol stream (QPSK, r = 1/2 k = 7, 1024-byte CADUs) ->
// frame correlator on the ENCODED sync word -> rotate_soft -> viterbi::Viterbi27 -> (NRZ-M) -> derandomiser -> RS(255,223) x 4 -> CADUs.
// the Viterbi27-based plugin decoders. Everything between the soft bytes and the CADUs stays in HBM:
//   k_lrpt_hard    thread per 32 soft bytes: the hard decisions Correlator::correlate packs (correlator.cpp:73-87), one bit array for the stream
//   k_lrpt_chain   ONE workgroup walks the frames in order: where frame k starts depends on where the correlator placed frame k-1
//                  (the module reads 16384 bytes, slides them by the correlator's offset and reads the rest: :217-227), so the walk is a
//                  pointer chase -- but each step is 8160 offsets x 8 sync-word variants of a 64-bit XOR + popcount, spread over the
//                  workgroup's 1024 lanes and reduced with the reference's own order of preference (first maximum: lowest offset, then lowest
//                  variant; the "> 45 at offset 0" shortcut first, :133-149). A few microseconds per frame; output = frame descriptors.
//   k_lrpt_gather  the descriptors' 16384 soft bytes each, turned the way rotate_soft does (rotation.cpp:4-58: -128 -> -127, swap, quarter turns)
//   Viterbi27      viterbi27_frames (fec_engine.hip): the packed lane-per-segment decoder, chained start states, Viterbi27::ber()
//   k_lrpt_post    NRZ-M (nrzm.cpp:13-22, the carried last bit is the previous frame's raw last bit), derand_ccsds from byte 4, the module's
//                  "frame came out inverted" test on byte 9 (:242-247), the sync marker the module writes itself (:255-257)
//   k_rs           reedsolomon::ReedSolomon::decode_interlaved(.., false, 4, ..) -- conventional basis, I = 4 (launch_rs_only)
//   launch_compact frames whose four codewords all decoded, in order
// Not reproduced: what the module does with the last, partly stale buffer at the end of a FILE (a short read leaves the previous frame's
// rotated bytes in the buffer's tail, and a file that ends exactly on a buffer is followed by one more iteration on the old buffer: usually
// the last CADU written twice). The engine keeps an incomplete frame for the next call and drops it at the end.
#include "../../include/sdhip.h"
#include "common.h"
#include "fec_kernels.h"

#include <algorithm>
#include <cstring>
#include <deque>
#include <vector>

namespace sdhip
{
    constexpr int LRPT_ENC = 16384;  // ENCODED_FRAME_SIZE
    constexpr int LRPT_FRAME = 1024; // FRAME_SIZE
    constexpr int LRPT_OFFSETS = (LRPT_ENC / 8 - 8) * 4; // symbol offsets the correlator scans (correlator.cpp:153-155)

    struct LrptDesc
    {
        long long start; // first soft byte of the frame in the stream
        int phase, swap, cor, locked;
    };
    struct LrptSync
    {
        unsigned long long w[8];
    };

    // bit (31 - (i & 31)) of word i >> 5 = soft[i] > 0
    __global__ __launch_bounds__(256) void k_lrpt_hard(const int8_t *__restrict__ soft, long long n, unsigned *bits, long long nwords)
    {
        const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
        if (w >= nwords)
            return;
        unsigned v = 0;
        const long long base = w * 32;
#pragma unroll 8
        for (int b = 0; b < 32; b++)
        {
            const long long i = base + b;
            const int s = i < n ? (int)soft[i] : 0;
            v |= (s > 0 ? 1u : 0u) << (31 - b);
        }
        bits[w] = v;
    }

    __device__ __forceinline__ unsigned long long lrpt_window(const unsigned *__restrict__ bits, long long j)
    { // stream bits [j, j + 64), first bit in bit 63
        const long long w = j >> 5;
        const unsigned sh = (unsigned)(j & 31);
        const unsigned a = bits[w], b = bits[w + 1], c = bits[w + 2];
        const unsigned hi = sh ? ((a << sh) | (b >> (32 - sh))) : a;
        const unsigned lo = sh ? ((b << sh) | (c >> (32 - sh))) : b;
        return ((unsigned long long)hi << 32) | lo;
    }

    // Correlator::correlate (QPSK) + the module's buffer bookkeeping, frame after frame
    __global__ __launch_bounds__(1024) void k_lrpt_chain(const unsigned *__restrict__ bits, long long n, LrptSync sync, LrptDesc *descs, int max_frames, int *count,
                                                          long long *consumed, long long o0, long long desc_base)
    {
        __shared__ unsigned red[16];
        __shared__ int s_p0, s_c0;
        const int tid = (int)threadIdx.x;
        long long o = o0; // (start inside the bit array handed in; descriptors name positions desc_base further on: the stream's)
        int cnt = 0;
        while (o + LRPT_ENC <= n && cnt < max_frames)
        {
            if (tid == 0)
            { // "Check pos 0": the first variant above 45 wins outright
                const unsigned long long win = lrpt_window(bits, o);
                int p0 = -1, c0 = 0;
                for (int p = 0; p < 8; p++)
                {
                    const int c = 64 - __popcll(sync.w[p] ^ win);
                    if (c > 45)
                    {
                        p0 = p;
                        c0 = c;
                        break;
                    }
                }
                s_p0 = p0;
                s_c0 = c0;
            }
            __syncthreads();
            long long pos = 0;
            int p = s_p0, cor = s_c0;
            if (p < 0)
            { // "Check the rest": the maximum over (offset ascending, variant ascending), the first one met
                unsigned best = 0;
                if (tid * 8 < LRPT_OFFSETS)
                {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                    {
                        const int s = tid * 8 + q;
                        const unsigned long long win = lrpt_window(bits, o + 2 * s);
#pragma unroll
                        for (int v = 0; v < 8; v++)
                        {
                            const unsigned c = 64u - (unsigned)__popcll(sync.w[v] ^ win);
                            const unsigned key = (c << 16) | ((unsigned)(8191 - s) << 3) | (unsigned)(7 - v);
                            best = best > key ? best : key;
                        }
                    }
                }
                for (int d = 32; d >= 1; d >>= 1)
                {
                    const unsigned other = (unsigned)__shfl_xor((int)best, d);
                    best = best > other ? best : other;
                }
                if ((tid & 63) == 0)
                    red[tid >> 6] = best;
                __syncthreads();
                unsigned all = red[0];
                for (int w = 1; w < 16; w++)
                    all = all > red[w] ? all : red[w];
                cor = (int)(all >> 16);
                const int s = 8191 - (int)((all >> 3) & 8191u);
                p = 7 - (int)(all & 7u);
                pos = 2 * s;
            }
            if (pos != 0 && o + pos + LRPT_ENC > n)
                break; // the slid frame is not complete yet: it stays for the next call
            if (tid == 0)
                descs[cnt] = LrptDesc{desc_base + o + pos, p % 4, (p / 4) == 0 ? 1 : 0, cor, pos == 0 ? 1 : 0};
            cnt++;
            o += pos + LRPT_ENC;
            __syncthreads(); // red / s_p0 are rewritten by the next frame
        }
        if (tid == 0)
        {
            *count = cnt;
            *consumed = o;
        }
    }

    // ---- the chain in parallel (round 5) -------------------------------------------------------------------------------------------------------------------
    // While the correlator answers "offset 0" -- the in-sync case, the "> 45 at offset 0" shortcut -- frame k starts exactly 16384 bytes behind frame k-1: the
    // chain runs along a GRID from wherever it last slid to. k_lrpt_spec takes the correlator's decision at EVERY grid position behind an origin at once (a
    // workgroup per position: the shortcut, else the full 8160 x 8 scan with the reference's order of preference); k_lrpt_walk then follows the chain over those
    // answers for as long as they say 0, takes the first slide (the module reads on behind it: a new origin) and stops there. In lock one round covers the call;
    // every slide costs a round, and a stream that keeps sliding (noise) goes back to the serial kernel above after a few of them. Same descriptors either way.
    struct LrptSpec
    {
        int pos, p, cor;
    };
    __global__ __launch_bounds__(256) void k_lrpt_spec(const unsigned *__restrict__ bits, long long n, long long origin, LrptSync sync, LrptSpec *spec, int npos)
    {
        __shared__ unsigned red[4];
        __shared__ int s_p0, s_c0;
        const int k = (int)blockIdx.x, tid = (int)threadIdx.x;
        if (k >= npos)
            return;
        const long long o = origin + (long long)k * LRPT_ENC;
        if (tid == 0)
        {
            const unsigned long long win = lrpt_window(bits, o);
            int p0 = -1, c0 = 0;
            for (int p = 0; p < 8; p++)
            {
                const int c = 64 - __popcll(sync.w[p] ^ win);
                if (c > 45)
                {
                    p0 = p;
                    c0 = c;
                    break;
                }
            }
            s_p0 = p0;
            s_c0 = c0;
        }
        __syncthreads();
        if (s_p0 >= 0)
        {
            if (tid == 0)
                spec[k] = LrptSpec{0, s_p0, s_c0};
            return;
        }
        unsigned best = 0;
        for (int s = tid; s < LRPT_OFFSETS; s += 256)
        {
            const unsigned long long win = lrpt_window(bits, o + 2 * s);
#pragma unroll
            for (int v = 0; v < 8; v++)
            {
                const unsigned c = 64u - (unsigned)__popcll(sync.w[v] ^ win);
                const unsigned key = (c << 16) | ((unsigned)(8191 - s) << 3) | (unsigned)(7 - v);
                best = best > key ? best : key;
            }
        }
        for (int d = 32; d >= 1; d >>= 1)
        {
            const unsigned other = (unsigned)__shfl_xor((int)best, d);
            best = best > other ? best : other;
        }
        if ((tid & 63) == 0)
            red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0)
        {
            unsigned all = red[0];
            for (int w = 1; w < 4; w++)
                all = all > red[w] ? all : red[w];
            spec[k] = LrptSpec{2 * (8191 - (int)((all >> 3) & 8191u)), 7 - (int)(all & 7u), (int)(all >> 16)};
        }
    }
    // One workgroup: the module's loop over the speculated answers (k_lrpt_chain's bookkeeping). A run of "offset 0" answers is a run of frames on the grid: the
    // first answer that is not 0 is found by the whole workgroup, the descriptors in front of it are written in parallel, thread 0 takes the slide. state[0] =
    // frames so far, state[1] = 1 when the stream is used up (or the slid frame is not complete yet), consumed = where the chain stands
    __global__ __launch_bounds__(1024) void k_lrpt_walk(const LrptSpec *__restrict__ spec, int npos, long long n, long long origin, LrptDesc *descs, int max_frames, int *state,
                                                         long long *consumed)
    {
        __shared__ int s_first;
        const int tid = (int)threadIdx.x;
        const int cnt0 = state[0];
        long long fit = (n - origin) / LRPT_ENC; // grid positions whose frame lies inside the stream
        int limit = npos;
        if (fit < (long long)limit)
            limit = (int)(fit < 0 ? 0 : fit);
        if (max_frames - cnt0 < limit)
            limit = max_frames - cnt0 < 0 ? 0 : max_frames - cnt0;
        if (tid == 0)
            s_first = limit;
        __syncthreads();
        for (int k = tid; k < limit; k += 1024)
            if (spec[k].pos != 0)
                atomicMin(&s_first, k);
        __syncthreads();
        const int first = s_first;
        for (int k = tid; k < first; k += 1024)
        {
            const LrptSpec sp = spec[k];
            descs[cnt0 + k] = LrptDesc{origin + (long long)k * LRPT_ENC, sp.p % 4, (sp.p / 4) == 0 ? 1 : 0, sp.cor, 1};
        }
        if (tid == 0)
        {
            int cnt = cnt0 + first, done = 1;
            long long o = origin + (long long)first * LRPT_ENC;
            if (first < limit)
            { // the first slide: the frame behind it, then off the grid -- the next round speculates from there
                const LrptSpec sp = spec[first];
                if (o + sp.pos + LRPT_ENC <= n)
                {
                    descs[cnt] = LrptDesc{o + sp.pos, sp.p % 4, (sp.p / 4) == 0 ? 1 : 0, sp.cor, 0};
                    cnt++;
                    o += sp.pos + LRPT_ENC;
                    done = (o + LRPT_ENC <= n && cnt < max_frames) ? 0 : 1;
                } // (else: the slid frame is not complete yet -- it stays for the next call)
            }
            state[0] = cnt;
            state[1] = done;
            *consumed = o;
        }
    }

    // rotate_soft(buffer, 16384, phase, swap), rotation.cpp:4-58, out of place: thread per (I, Q) pair
    __global__ __launch_bounds__(256) void k_lrpt_gather(const int8_t *__restrict__ soft, const LrptDesc *__restrict__ descs, int nframes, int8_t *frames)
    {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x; // pair index over all frames
        const int f = (int)(i / (LRPT_ENC / 2));
        if (f >= nframes)
            return;
        const int k = (int)(i - (long long)f * (LRPT_ENC / 2));
        const LrptDesc d = descs[f];
        const char2 in = *reinterpret_cast<const char2 *>(soft + d.start + 2 * k);
        int a = in.x, b = in.y;
        a = a == -128 ? -127 : a;
        b = b == -128 ? -127 : b;
        if (d.swap)
        {
            const int t = a;
            a = b;
            b = t;
        }
        if (d.phase == 1)
        {
            const int t = a;
            a = b;
            b = -t;
        }
        else if (d.phase == 2)
        {
            a = -a;
            b = -b;
        }
        else if (d.phase == 3)
        {
            const int t = a;
            a = -b;
            b = t;
        }
        char2 o;
        o.x = (signed char)a;
        o.y = (signed char)b;
        *reinterpret_cast<char2 *>(frames + (size_t)f * LRPT_ENC + 2 * k) = o;
    }

    // diff.decode -> derand_ccsds(&frame[4], 1020) -> inverted-frame test -> sync marker; thread per output byte
    __global__ __launch_bounds__(256) void k_lrpt_post(const unsigned char *__restrict__ raw, int nframes, int diff, int carry_bit, const unsigned char *__restrict__ pn,
                                                       unsigned char *out)
    {
        const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
        const int f = (int)(g / LRPT_FRAME);
        if (f >= nframes)
            return;
        const int i = (int)(g % LRPT_FRAME);
        const unsigned char *r = raw + (size_t)f * LRPT_FRAME;
        auto byte_at = [&](int k) -> unsigned {
            unsigned b = r[k];
            if (diff)
            {
                const unsigned prev = k > 0 ? (unsigned)(r[k - 1] & 1) : (f > 0 ? (unsigned)(r[-1] & 1) : (unsigned)carry_bit);
                b ^= ((b >> 1) & 0x7Fu) | (prev << 7);
            }
            if (k >= 4)
                b ^= pn[(k - 4) % 255];
            return b & 0xFFu;
        };
        unsigned v = byte_at(i);
        if (byte_at(9) == 0xFFu) // "There is a VERY rare edge case where CADUs end up inverted"
            v ^= 0xFFu;
        if (i < 4)
            v = i == 0 ? 0x1Du : (i == 1 ? 0xCFu : (i == 2 ? 0xFCu : 0x1Du));
        out[g] = (unsigned char)v;
    }

    // rotate_64 / swapIQ / the constructor's eight variants, correlator.cpp:3-66
    static unsigned long long lrpt_rotate_64(unsigned long long word, int p)
    {
        const unsigned long long i = word & 0xaaaaaaaaaaaaaaaaull, q = word & 0x5555555555555555ull;
        switch (p)
        {
        case 1:
            word = ((i ^ 0xaaaaaaaaaaaaaaaaull) >> 1) | (q << 1);
            break;
        case 2:
            word = word ^ 0xffffffffffffffffull;
            break;
        case 3:
            word = (i >> 1) | ((q ^ 0x5555555555555555ull) << 1);
            break;
        default:
            break;
        }
        return ((word & 0x5555555555555555ull) << 1) | ((word & 0xAAAAAAAAAAAAAAAAull) >> 1);
    }
    static unsigned long long lrpt_swap_iq(unsigned long long in)
    {
        const unsigned long long i = in & 0xaaaaaaaaaaaaaaaaull, q = in & 0x5555555555555555ull;
        return (i >> 1) | (q << 1);
    }

    struct LrptDecoder
    {
        sdhip_lrpt_cfg cfg;
        sdhip_lrpt_stats stats{};
        LrptSync sync;
        DevBuf<int8_t> stream; // the soft bytes not yet consumed (carry) followed by the call's
        size_t carry = 0;
        DevBuf<unsigned> bits;
        DevBuf<LrptDesc> d_desc;
        DevBuf<int> d_cnt;
        DevBuf<long long> d_consumed;
        DevBuf<LrptSpec> d_spec;
        DevBuf<int> d_state;
        unsigned long long stats_rounds = 0;
        unsigned vit_enc = 0; // Viterbi27's BER re-encoder register, carried across calls
        DevBuf<int8_t> d_frames;
        DevBuf<unsigned char> d_raw, d_post, d_pn, d_clean;
        DevBuf<int> d_err, d_dst;
        int vit_start = -2; // Viterbi27's CCDecoder: unbiased on its first call, chained afterwards
        int nrzm_last = 0;
        std::deque<std::vector<unsigned char>> outq; // host path
        std::vector<int8_t> pend;

        explicit LrptDecoder(const sdhip_lrpt_cfg &c) : cfg(c)
        {
            SD_HIP(hipSetDevice(cfg.device));
            // module_meteor_lrpt_decoder.cpp:204: the encoded sync word, plain or behind NRZ-M
            const unsigned long long sw = cfg.diff_decode ? 0xfc4ef4fd0cc2df89ull : 0xfca2b63db00d9794ull;
            for (int i = 0; i < 4; i++)
                sync.w[i] = lrpt_rotate_64(sw, i);
            for (int i = 4; i < 8; i++)
                sync.w[i] = lrpt_rotate_64(lrpt_swap_iq(sw) ^ 0xFFFFFFFFFFFFFFFFull, i - 4);
            // the CCSDS pseudo-noise sequence (randomization.cpp's ccsds_pn: h(x) = x^8 + x^7 + x^5 + x^3 + 1 from all ones), 255 bytes
            unsigned char pn[255];
            unsigned reg = 0xFF;
            for (int i = 0; i < 255; i++)
            {
                unsigned b = 0;
                for (int k = 0; k < 8; k++)
                {
                    const unsigned o = reg & 1u; // output = x^0 end of the register
                    b = (b << 1) | o;
                    const unsigned fb = ((reg >> 0) ^ (reg >> 3) ^ (reg >> 5) ^ (reg >> 7)) & 1u;
                    reg = (reg >> 1) | (fb << 7);
                }
                pn[i] = (unsigned char)b;
            }
            d_pn.reserve(256);
            SD_HIP(hipMemcpy(d_pn.p, pn, 255, hipMemcpyHostToDevice));
            d_cnt.reserve(1);
            d_consumed.reserve(1);
            stats.viterbi_ber = 10.0f;
            for (int i = 0; i < 4; i++)
                stats.rs_errors[i] = -1;
        }

        // d_in: n new soft bytes (device). CADUs to d_out (device, cap_frames frames). Returns the frames written.
        int64_t process(const int8_t *d_in, size_t n, unsigned char *d_out, size_t cap_frames)
        {
            SD_HIP(hipSetDevice(cfg.device));
            stats.soft_in += n;
            const size_t total = carry + n;
            // with nothing pending the call's own buffer is the stream (no copy); else the new bytes are appended to the pending ones
            const int8_t *base = d_in;
            if (carry || total < (size_t)LRPT_ENC || (reinterpret_cast<uintptr_t>(d_in) & 1u)) // (the gather reads (I, Q) pairs as one 2-byte load)
            {
                if (total + 64 > stream.cap)
                { // (grow keeping the pending bytes)
                    DevBuf<int8_t> bigger;
                    bigger.reserve(total + total / 2 + 64);
                    if (carry)
                        SD_HIP(hipMemcpy(bigger.p, stream.p, carry, hipMemcpyDeviceToDevice));
                    stream.swap(bigger);
                }
                if (n)
                    SD_HIP(hipMemcpy(stream.p + carry, d_in, n, hipMemcpyDeviceToDevice));
                base = stream.p;
            }
            if (total < (size_t)LRPT_ENC)
            {
                carry = total;
                return 0;
            }
            const long long nwords = (long long)((total + 31) / 32);
            bits.reserve((size_t)nwords + 4);
            SD_HIP(hipMemsetAsync(bits.p + nwords, 0, 4 * sizeof(unsigned), nullptr));
            hipLaunchKernelGGL(k_lrpt_hard, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, nullptr, base, (long long)total, bits.p, nwords);
            const int max_frames = (int)(total / LRPT_ENC) + 1;
            d_desc.reserve(max_frames);
            int nf = 0;
            long long consumed = 0;
            static const bool serial_only = getenv("SDHIP_LRPT_SERIAL") && atoi(getenv("SDHIP_LRPT_SERIAL")) != 0;
            bool chained = false;
            if (!serial_only)
            { // the chain along its grid, a round per slide (k_lrpt_spec / k_lrpt_walk)
                ProfScope _ps("k_lrpt_spec + k_lrpt_walk", nullptr);
                d_spec.reserve(max_frames);
                d_state.reserve(2);
                SD_HIP(hipMemsetAsync(d_state.p, 0, 2 * sizeof(int), nullptr));
                long long origin = 0;
                int st[2] = {0, 0};
                for (int round = 0; round < 6; round++)
                {
                    const int npos = (int)(((long long)total - origin) / LRPT_ENC);
                    if (npos <= 0)
                    {
                        st[1] = 1;
                        consumed = origin;
                        break;
                    }
                    hipLaunchKernelGGL(k_lrpt_spec, dim3((unsigned)npos), dim3(256), 0, nullptr, bits.p, (long long)total, origin, sync, d_spec.p, npos);
                    hipLaunchKernelGGL(k_lrpt_walk, dim3(1), dim3(1024), 0, nullptr, d_spec.p, npos, (long long)total, origin, d_desc.p, max_frames, d_state.p, d_consumed.p);
                    SD_HIP(hipMemcpy(st, d_state.p, sizeof(st), hipMemcpyDeviceToHost));
                    SD_HIP(hipMemcpy(&consumed, d_consumed.p, sizeof(long long), hipMemcpyDeviceToHost));
                    origin = consumed;
                    if (st[1])
                        break;
                }
                nf = st[0];
                chained = st[1] != 0;
                stats_rounds++;
            }
            if (!chained)
            { // a stream that keeps sliding (or SDHIP_LRPT_SERIAL=1): the serial walk over what is left, behind the frames found so far
                ProfScope _ps("k_lrpt_chain", nullptr);
                long long rest_consumed = 0;
                int rest = 0;
                const long long wofs = consumed / 32 * 32; // (the serial kernel addresses the bit array from a word boundary: it gets the sub-array and the byte offset inside its first word)
                hipLaunchKernelGGL(k_lrpt_chain, dim3(1), dim3(1024), 0, nullptr, bits.p + wofs / 32, (long long)total - wofs, sync, d_desc.p + nf, max_frames - nf, d_cnt.p, d_consumed.p,
                                   consumed - wofs, wofs);
                SD_HIP(hipMemcpy(&rest, d_cnt.p, sizeof(int), hipMemcpyDeviceToHost));
                SD_HIP(hipMemcpy(&rest_consumed, d_consumed.p, sizeof(long long), hipMemcpyDeviceToHost));
                nf += rest;
                consumed = wofs + rest_consumed;
            }
            int64_t written = 0;
            if (nf > 0)
            {
                d_frames.reserve((size_t)nf * LRPT_ENC);
                d_raw.reserve((size_t)nf * LRPT_FRAME + 64);
                d_post.reserve((size_t)nf * LRPT_FRAME + 64);
                d_err.reserve((size_t)nf * 4);
                d_dst.reserve(nf);
                const long long pairs = (long long)nf * (LRPT_ENC / 2);
                hipLaunchKernelGGL(k_lrpt_gather, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, nullptr, base, d_desc.p, nf, d_frames.p);
                std::vector<int> ber;
                int ret = vit_start;
                viterbi27_frames(LRPT_ENC / 2, 1024, d_frames.p, nf, vit_start, d_raw.p, &ber, &ret, &vit_enc); // Viterbi27(ENCODED_FRAME_SIZE / 2, polys): ber_test_size 1024
                vit_start = ret;
                const long long nb = (long long)nf * LRPT_FRAME;
                hipLaunchKernelGGL(k_lrpt_post, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, nullptr, d_raw.p, nf, cfg.diff_decode ? 1 : 0, nrzm_last, d_pn.p, d_post.p);
                unsigned char lastb = 0;
                SD_HIP(hipMemcpy(&lastb, d_raw.p + (size_t)nf * LRPT_FRAME - 1, 1, hipMemcpyDeviceToHost));
                nrzm_last = lastb & 1;
                d_clean.reserve(rs_scratch_bytes((long long)nf * 4));
                launch_rs_only(d_post.p + 4, nf, LRPT_FRAME, 0, 4, 32, 0, d_err.p, nullptr, d_clean.p); // ReedSolomon(RS223): fill_bytes 0 (reedsolomon.h:29)
                std::vector<int> err((size_t)nf * 4), dst(nf);
                SD_HIP(hipMemcpy(err.data(), d_err.p, err.size() * sizeof(int), hipMemcpyDeviceToHost));
                int keep = 0;
                for (int f = 0; f < nf; f++)
                {
                    const bool ok = err[4 * f] >= 0 && err[4 * f + 1] >= 0 && err[4 * f + 2] >= 0 && err[4 * f + 3] >= 0;
                    dst[f] = ok ? keep++ : -1;
                }
                if ((size_t)keep > cap_frames)
                    throw HipError("lrpt: CADU output buffer too small");
                SD_HIP(hipMemcpy(d_dst.p, dst.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice));
                if (keep)
                    launch_compact(d_post.p, d_dst.p, nf, LRPT_FRAME, d_out, nullptr);
                SD_HIP(hipDeviceSynchronize());
                written = keep;
                LrptDesc last;
                SD_HIP(hipMemcpy(&last, d_desc.p + (nf - 1), sizeof(last), hipMemcpyDeviceToHost));
                stats.frames_seen += nf;
                stats.frames_out += keep;
                stats.correlator_lock = last.locked;
                stats.cor = last.cor;
                stats.viterbi_ber = ((float)ber[nf - 1] / 1024.0f) * 4.0f;
                for (int i = 0; i < 4; i++)
                    stats.rs_errors[i] = err[4 * (size_t)(nf - 1) + i];
            }
            // keep what the walk did not consume
            const size_t rest = total - (size_t)consumed;
            if (rest && (consumed || base != stream.p))
            {
                if (base == stream.p)
                {
                    DevBuf<int8_t> tmp;
                    tmp.reserve(rest);
                    SD_HIP(hipMemcpy(tmp.p, stream.p + consumed, rest, hipMemcpyDeviceToDevice));
                    SD_HIP(hipMemcpy(stream.p, tmp.p, rest, hipMemcpyDeviceToDevice));
                }
                else
                {
                    stream.reserve(rest + 64);
                    SD_HIP(hipMemcpy(stream.p, base + consumed, rest, hipMemcpyDeviceToDevice));
                }
            }
            carry = rest;
            return written;
        }
    };
} // namespace sdhip

using namespace sdhip;

#define SD_GUARD_BEGIN try {
#define SD_GUARD_END(ret)                \
    }                                    \
    catch (const std::exception &e)      \
    {                                    \
        sdhip::set_error(e.what());      \
        return ret;                      \
    }

extern "C"
{
    void sdhip_lrpt_cfg_default(sdhip_lrpt_cfg *c)
    {
        memset(c, 0, sizeof(*c));
        c->diff_decode = 0;
        c->device = 0;
    }
    void *sdhip_lrpt_create(const sdhip_lrpt_cfg *c)
    {
        SD_GUARD_BEGIN
        return new LrptDecoder(*c);
        SD_GUARD_END(nullptr)
    }
    void sdhip_lrpt_destroy(void *h) { delete (LrptDecoder *)h; }
    int64_t sdhip_lrpt_process_dev(void *h, const int8_t *d_soft, size_t n, uint8_t *d_cadu, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        return ((LrptDecoder *)h)->process(d_soft, n, d_cadu, cap_frames);
        SD_GUARD_END(-1)
    }
    int sdhip_lrpt_push(void *h, const int8_t *soft, size_t n)
    {
        SD_GUARD_BEGIN
        LrptDecoder *e = (LrptDecoder *)h;
        SD_HIP(hipSetDevice(e->cfg.device));
        if (n == 0)
            return 0;
        DevBuf<int8_t> din;
        DevBuf<unsigned char> dout;
        const size_t cap = (e->carry + n) / LRPT_ENC + 2;
        din.reserve(n);
        dout.reserve(cap * LRPT_FRAME);
        SD_HIP(hipMemcpy(din.p, soft, n, hipMemcpyHostToDevice));
        const int64_t k = e->process(din.p, n, dout.p, cap);
        if (k > 0)
        {
            std::vector<unsigned char> v((size_t)k * LRPT_FRAME);
            SD_HIP(hipMemcpy(v.data(), dout.p, v.size(), hipMemcpyDeviceToHost));
            e->outq.push_back(std::move(v));
        }
        return 0;
        SD_GUARD_END(-1)
    }
    int64_t sdhip_lrpt_pull(void *h, uint8_t *cadu, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        LrptDecoder *e = (LrptDecoder *)h;
        size_t got = 0;
        while (got < cap_frames && !e->outq.empty())
        {
            std::vector<unsigned char> &v = e->outq.front();
            const size_t have = v.size() / LRPT_FRAME, take = std::min(have, cap_frames - got);
            memcpy(cadu + got * LRPT_FRAME, v.data(), take * LRPT_FRAME);
            got += take;
            if (take == have)
                e->outq.pop_front();
            else
                v.erase(v.begin(), v.begin() + take * LRPT_FRAME);
        }
        return (int64_t)got;
        SD_GUARD_END(-1)
    }
    int sdhip_lrpt_get_stats(void *h, sdhip_lrpt_stats *st)
    {
        SD_GUARD_BEGIN
        *st = ((LrptDecoder *)h)->stats;
        return 0;
        SD_GUARD_END(-1)
    }
}
