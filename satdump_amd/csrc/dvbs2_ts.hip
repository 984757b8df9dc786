// dvbs2_ts.hip -- the step behind the DVB-S2 BBFRAMEs (SURVEY.md 8 f-2's tail, VERDICT r5 missing 3): MPEG transport-stream packets out of the BBFRAMEs'
// data fields, dvbs2::BBFrameTSParser::work (src-core/common/codings/dvb-s2/bbframe_ts_parser.cpp:96-243) as plugins/dvb_support/dvbs2/
// module_s2_ts_extractor.cpp:77-105 calls it: ONE frame per call, the parser's state carried from frame to frame.
//
// The parser is a byte-serial state machine, but what it decides depends on the ten header bytes of every frame only (CRC-8 over the header, DFL, SYNCD) and on
// counters: which bytes of which data field form which 188-byte packet is arithmetic. So the HOST walks the headers (80 bytes per 7 KB frame cross PCIe) with the
// parser's own state variables -- count, index, spanning, distance, synched, statement for statement at packet granularity -- and writes one descriptor per packet
// the parser would emit; the DEVICE gathers the packets (a thread per packet: 187 payload bytes from one or more data fields, the sync byte 0x47 in front), runs the
// CRC-8 the modulator put in place of the next sync byte (EN 302 307 5.1.4: polynomial 0xD5, bbframe_ts_parser.cpp:50-75) and sets the transport error indicator
// exactly where the parser does. Data fields stay in HBM (the BBFRAMEs the demodulator module left there); the packet in flight between two calls is carried in a
// 376-byte device buffer.
//
// Quirks of the parser that are part of its output and reproduced: (1) a packet that ends exactly with its data field (its CRC byte is the first byte of the next
// frame) never gets its error indicator -- the parser ORs it into tsframes[tei_p] with tei_p reset to 0 by the new call (:106,203), a byte the next packet's sync
// byte then overwrites; (2) a spanning packet that is complete but not yet written when the parser loses sync (header CRC, DFL, SYNCD mismatch) is dropped (:152-156);
// (3) a SYNCD mismatch takes effect from the NEXT frame (:187-193). Not reproduced: a resynchronisation whose SYNCD points behind the data field -- the parser's
// unsigned df_remaining wraps and it reads 4 G bytes past the frame (:146-147); here such a frame is skipped and the parser stays out of sync.
#include "common.h"
#include "../../include/sdhip.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace sdhip
{
    struct TsFrameRef // where a frame's consumed data-field bytes sit in the virtual byte stream of the call
    {
        long long base; // stream position of the first byte
        int frame;      // index into the call's frames; -1 = the carry buffer
        int offset;     // byte offset inside the frame (the carry buffer)
        int len;
        int pad;
    };
    struct TsPacket
    {
        long long pos; // stream position of the packet's first payload byte (187 of them, then the CRC byte)
        int tei_check; // 1: compare the CRC byte (it lies inside the stream) and set the error indicator on a mismatch
        int pad;
    };

    __device__ __forceinline__ int ts_find(const TsFrameRef *refs, int nrefs, long long pos)
    { // the last ref with base <= pos
        int lo = 0, hi = nrefs - 1;
        while (lo < hi)
        {
            const int mid = (lo + hi + 1) >> 1;
            if (refs[mid].base <= pos)
                lo = mid;
            else
                hi = mid - 1;
        }
        return lo;
    }
    __device__ __forceinline__ const unsigned char *ts_src(const TsFrameRef &r, const unsigned char *frames, int frame_bytes, const unsigned char *carry)
    {
        return r.frame < 0 ? carry + r.offset : frames + (size_t)r.frame * frame_bytes + r.offset;
    }

    __global__ __launch_bounds__(64) void k_ts_headers(const unsigned char *frames, int frame_bytes, int nframes, unsigned char *hdr)
    {
        const int k = (int)(blockIdx.x * 64 + threadIdx.x);
        if (k >= nframes)
            return;
        for (int b = 0; b < 10; b++)
            hdr[(size_t)k * 10 + b] = frames[(size_t)k * frame_bytes + b];
    }

    // thread per packet (crc_tab in LDS: the parser's table, bbframe_ts_parser.cpp:50-75)
    __global__ __launch_bounds__(256) void k_ts_extract(const unsigned char *frames, int frame_bytes, const unsigned char *carry, const TsFrameRef *refs, int nrefs,
                                                         const TsPacket *pk, int npk, unsigned char *out)
    {
        __shared__ unsigned char tab[256];
        {
            const int i = (int)threadIdx.x;
            int r = i, crc = 0;
            for (int j = 7; j >= 0; j--)
            {
                if (((r & (1 << j)) ? 1 : 0) ^ ((crc & 0x80) ? 1 : 0))
                    crc = (crc << 1) ^ 0xD5;
                else
                    crc <<= 1;
            }
            tab[i] = (unsigned char)crc;
        }
        __syncthreads();
        const int j = (int)(blockIdx.x * 256 + threadIdx.x);
        if (j >= npk)
            return;
        const TsPacket p = pk[j];
        unsigned char *o = out + (size_t)j * 188;
        int ri = ts_find(refs, nrefs, p.pos);
        TsFrameRef r = refs[ri];
        const unsigned char *src = ts_src(r, frames, frame_bytes, carry);
        long long at = p.pos;
        unsigned char crc = 0, b1 = 0;
        o[0] = 0x47;
        for (int i = 0; i < 187 + (p.tei_check ? 1 : 0); i++, at++)
        {
            while (at >= r.base + r.len)
            {
                r = refs[++ri];
                src = ts_src(r, frames, frame_bytes, carry);
            }
            const unsigned char v = src[at - r.base];
            if (i < 187)
            {
                crc = tab[v ^ crc];
                if (i == 0)
                    b1 = v;
                else
                    o[1 + i] = v;
            }
            else if (v != crc)
                b1 |= 0x80; // TS_ERROR_INDICATOR
        }
        o[1] = b1;
    }

    // the bytes of the stream from `from` on, into the carry buffer (the packet in flight at the end of a call)
    __global__ __launch_bounds__(64) void k_ts_carry(const unsigned char *frames, int frame_bytes, const unsigned char *carry_in, const TsFrameRef *refs, int nrefs, long long from, int n,
                                                      unsigned char *carry_out)
    {
        const int i = (int)threadIdx.x + 64 * (int)blockIdx.x;
        if (i >= n)
            return;
        const long long at = from + i;
        const int ri = ts_find(refs, nrefs, at);
        const TsFrameRef r = refs[ri];
        carry_out[i] = ts_src(r, frames, frame_bytes, carry_in)[at - r.base];
    }

    struct TsEngine
    {
        int device = 0, kbch = 0, frame_bytes = 0;
        unsigned max_dfl = 0;
        hipStream_t stream = nullptr;
        // BBFrameTSParser's members (bbframe_ts_parser.h:66-77) -- what of them is not data
        unsigned count = 0, synched = 0, distance = 0, spanning = 0, index = 0;
        // the packet in flight: `have` bytes of its stream range [its first payload byte, ...) sit in carry[sel]
        int carry_have = 0, carry_sel = 0;
        DevBuf<unsigned char> d_carry[2], d_hdr, d_stage, d_out;
        DevBuf<TsFrameRef> d_refs;
        DevBuf<TsPacket> d_pk;
        std::vector<unsigned char> h_hdr;
        unsigned char crc_tab[256];
        uint64_t frames_in = 0, packets_out = 0, header_crc_fails = 0, resyncs = 0;

        explicit TsEngine(int dev, int bbframe_bits) : device(dev), kbch(bbframe_bits), frame_bytes(bbframe_bits / 8)
        {
            if (bbframe_bits < 80 + 188 * 8 || bbframe_bits % 8)
                throw HipError("dvbs2 ts: bbframe size must be a whole number of bytes and hold a header and a packet");
            max_dfl = (unsigned)kbch - 80;
            SD_HIP(hipSetDevice(device));
            SD_HIP(hipStreamCreate(&stream));
            d_carry[0].reserve(512);
            d_carry[1].reserve(512);
            for (int i = 0; i < 256; i++)
            {
                int r = i, crc = 0;
                for (int j = 7; j >= 0; j--)
                {
                    if (((r & (1 << j)) ? 1 : 0) ^ ((crc & 0x80) ? 1 : 0))
                        crc = (crc << 1) ^ 0xD5;
                    else
                        crc <<= 1;
                }
                crc_tab[i] = (unsigned char)crc;
            }
        }
        ~TsEngine()
        {
            if (stream)
                (void)hipStreamDestroy(stream);
        }
        // check_crc8 over the 80 header bits (bbframe_ts_parser.cpp:82-94)
        static unsigned header_crc(const unsigned char *in)
        {
            int crc = 0;
            for (int n = 0; n < 80; n++)
            {
                const int b = ((in[n / 8] >> (7 - (n % 8))) & 1) ^ (crc & 0x01);
                crc >>= 1;
                if (b)
                    crc ^= 0xAB;
            }
            return (unsigned)crc;
        }

        // frames resident on the device; packets to d_ts (device, cap packets). Returns the packets written.
        int64_t process_dev(const unsigned char *d_frames, int nframes, unsigned char *d_ts, size_t cap_packets)
        {
            SD_HIP(hipSetDevice(device));
            if (nframes <= 0)
                return 0;
            d_hdr.reserve((size_t)nframes * 10);
            h_hdr.resize((size_t)nframes * 10);
            hipLaunchKernelGGL(k_ts_headers, dim3((nframes + 63) / 64), dim3(64), 0, stream, d_frames, frame_bytes, nframes, d_hdr.p);
            SD_HIP(hipMemcpyAsync(h_hdr.data(), d_hdr.p, h_hdr.size(), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));

            // ---- the parser's walk, header by header (work(bbf, 1, ..) per frame: in_p / out_p / tei_p start at 0 in every frame). The VIRTUAL STREAM of the call =
            // the carried bytes of the spanning packet in flight (if any), then every data-field byte the parser consumes, in order.
            std::vector<TsFrameRef> refs;
            std::vector<TsPacket> pk;
            long long vend = 0;
            long long span_pos = 0; // stream position of the spanning packet in `packet` (valid while index > 0)
            if (index > 0 && carry_have > 0)
            {
                refs.push_back(TsFrameRef{0, -1, 0, carry_have, 0});
                vend = carry_have;
            }
            for (int k = 0; k < nframes; k++)
            {
                const unsigned char *h = h_hdr.data() + (size_t)k * 10;
                frames_in++;
                if (header_crc(h) != 0)
                { // :113-122
                    synched = 0;
                    header_crc_fails++;
                    continue;
                }
                const unsigned dfl = ((unsigned)h[4] << 8) | h[5], syncd = ((unsigned)h[7] << 8) | h[8];
                if (dfl > max_dfl || dfl % 8 != 0)
                { // :128-140
                    synched = 0;
                    continue;
                }
                long long D = dfl / 8;
                int pos = 10;
                if (!synched)
                { // :146-157: whatever was in flight is gone, a completed spanning packet too
                    const long long skip = (long long)(syncd / 8) + 1;
                    if (skip > D)
                        continue; // (the parser's unsigned df_remaining wraps here: see the file header)
                    pos += (int)skip;
                    D -= skip;
                    count = 0;
                    synched = 1;
                    index = 0;
                    spanning = 0;
                    distance = syncd / 8;
                    resyncs++;
                }
                if (D > 0)
                    refs.push_back(TsFrameRef{vend, k, pos, (int)D, 0});
                long long at = vend; // stream position of the next byte the parser consumes
                vend += D;
                bool first = true;  // crc_check, :185-191
                long long direct_pos = -1; // the direct packet started in THIS frame whose CRC byte has not gone by yet (tei_p is its own)
                while (D > 0)
                {
                    if (count == 0)
                    { // a packet starts (:161-192)
                        if (index == 188)
                        { // the completed spanning packet is written now; its CRC byte went by before this (count had to return to 0)
                            pk.push_back(TsPacket{span_pos, 1, 0});
                            index = 0;
                            spanning = 0;
                        }
                        if (D < 187)
                        {
                            index = 1;
                            spanning = 1;
                            span_pos = at;
                        }
                        else
                        {
                            pk.push_back(TsPacket{at, 0, 0}); // written as it is read; tei_check is set if its CRC byte comes within this frame
                            direct_pos = at;
                        }
                        count = 1;
                        if (first)
                        {
                            if (distance != syncd / 8)
                                synched = 0; // takes effect with the next frame
                            first = false;
                        }
                    }
                    else if (count == 188)
                    { // the CRC byte (:195-212)
                        if (!spanning && direct_pos >= 0)
                            pk.back().tei_check = 1; // (the direct packet of this frame is the last descriptor written: nothing is emitted between its start and here)
                        direct_pos = -1;
                        at++;
                        D--;
                        count = 0;
                        if (D == 0)
                            distance = 187;
                        continue;
                    }
                    const long long m = std::min<long long>(D, (long long)(188 - count)); // payload bytes up to the CRC byte or the end of the data field (:213-234)
                    at += m;
                    D -= m;
                    count += (unsigned)m;
                    if (spanning)
                        index += (unsigned)m;
                    distance = D == 0 ? 0 : distance + (unsigned)m;
                }
            }
            // ---- the packets, on the device
            const int npk = (int)pk.size();
            if ((size_t)npk > cap_packets)
                throw HipError("dvbs2 ts: output buffer too small");
            if (!refs.empty())
            {
                d_refs.reserve(refs.size());
                SD_HIP(hipMemcpyAsync(d_refs.p, refs.data(), refs.size() * sizeof(TsFrameRef), hipMemcpyHostToDevice, stream));
            }
            if (npk > 0)
            {
                d_pk.reserve(pk.size());
                SD_HIP(hipMemcpyAsync(d_pk.p, pk.data(), pk.size() * sizeof(TsPacket), hipMemcpyHostToDevice, stream));
                ProfScope _ps("k_ts_extract", stream);
                hipLaunchKernelGGL(k_ts_extract, dim3((npk + 255) / 256), dim3(256), 0, stream, d_frames, frame_bytes, d_carry[carry_sel].p, d_refs.p, (int)refs.size(), d_pk.p, npk, d_ts);
            }
            // ---- what the spanning packet in flight has consumed so far goes into the other carry buffer
            int keep = 0;
            if (index > 0)
            {
                keep = (int)(vend - span_pos);
                if (keep > 376)
                    throw HipError("dvbs2 ts: internal: carried packet longer than a packet");
                hipLaunchKernelGGL(k_ts_carry, dim3((keep + 63) / 64), dim3(64), 0, stream, d_frames, frame_bytes, d_carry[carry_sel].p, d_refs.p, (int)refs.size(), span_pos, keep,
                                   d_carry[carry_sel ^ 1].p);
            }
            SD_HIP(hipStreamSynchronize(stream)); // (refs / pk are locals)
            carry_sel ^= 1;
            carry_have = keep;
            packets_out += (uint64_t)npk;
            return npk;
        }
        // host buffers in and out
        int64_t process_host(const unsigned char *frames, int nframes, unsigned char *ts, size_t cap_packets)
        {
            SD_HIP(hipSetDevice(device));
            if (nframes <= 0)
                return 0;
            d_stage.reserve((size_t)nframes * frame_bytes);
            const size_t cap = std::min<size_t>(cap_packets, (size_t)nframes * (frame_bytes / 188 + 2));
            d_out.reserve(cap * 188 + 188);
            SD_HIP(hipMemcpyAsync(d_stage.p, frames, (size_t)nframes * frame_bytes, hipMemcpyHostToDevice, stream));
            const int64_t n = process_dev(d_stage.p, nframes, d_out.p, cap);
            if (n > 0)
            {
                SD_HIP(hipMemcpyAsync(ts, d_out.p, (size_t)n * 188, hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
            }
            return n;
        }
    };
} // namespace sdhip

using namespace sdhip;

#define SD_GUARD_BEGIN try {
#define SD_GUARD_END(ret)           \
    }                               \
    catch (const std::exception &e) \
    {                               \
        sdhip::set_error(e.what()); \
        return ret;                 \
    }

extern "C"
{
    void *sdhip_s2_ts_create(int device, int bbframe_bits)
    {
        SD_GUARD_BEGIN
        return (void *)new TsEngine(device, bbframe_bits);
        SD_GUARD_END(nullptr)
    }
    void sdhip_s2_ts_destroy(void *h) { delete (TsEngine *)h; }
    int64_t sdhip_s2_ts_process_dev(void *h, const uint8_t *d_bbframes, int nframes, uint8_t *d_ts, size_t cap_packets)
    {
        SD_GUARD_BEGIN
        return ((TsEngine *)h)->process_dev(d_bbframes, nframes, d_ts, cap_packets);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_s2_ts_process(void *h, const uint8_t *bbframes, int nframes, uint8_t *ts, size_t cap_packets)
    {
        SD_GUARD_BEGIN
        return ((TsEngine *)h)->process_host(bbframes, nframes, ts, cap_packets);
        SD_GUARD_END(-1)
    }
    int sdhip_s2_ts_get_stats(void *h, sdhip_s2_ts_stats *st)
    {
        if (!h || !st)
            return -1;
        const TsEngine *e = (const TsEngine *)h;
        st->frames_in = e->frames_in;
        st->packets_out = e->packets_out;
        st->header_crc_fails = e->header_crc_fails;
        st->resyncs = e->resyncs;
        st->synched = (int)e->synched;
        return 0;
    }
}
