// aos_demux.hip -- the first step behind the CADUs: CCSDS AOS virtual-channel handling as the reference's instrument decoders do it
// before anything else -- ccsds::ccsds_aos::parseVCDU (src-core/common/ccsds/ccsds_aos/vcdu.cpp:10-18) on every frame, the frames of one virtual channel
// picked out, and ccsds::ccsds_aos::Demuxer::work (demuxer.cpp:67-201, mpdu.cpp:10-14) turning their M_PDU zones into Space Packets -- with the CADUs and
// the packets' payload staying in HBM:
//   k_aos_vcdu      thread per frame: the VCDU primary header's fields                                          (data-parallel)
//   k_aos_select    stream compaction of the frames of one VCID, order kept                                     (data-parallel, block scan)
//   k_aos_scan      thread per frame: first header pointer, the chain of packet headers it starts (position + the six header bytes), the frame's first
//                   six zone bytes (a header split across two frames is completed from them)                     (data-parallel: a frame's chain does not
//                   depend on other frames)
//   host FSM        the Demuxer's state machine, frame after frame, on those descriptors alone -- which packet a byte range belongs to and in which
//                   frame's work() call a packet is handed out depends on carried state (a packet spanning frames, the "header takes priority" rule,
//                   the delayed hand-out of a packet that ends exactly on a zone's end): O(#packets) integer work, no payload byte crosses PCIe
//   k_aos_gather    the byte ranges the FSM decided on, copied into the packets' payload pool                    (data-parallel)
// Bit-exact with the reference's Demuxer including its quirks (the + 1 in the continuation length when a frame carries a header, frames whose pointer
// lies outside the zone skipped without touching the state, ...): tests/test_aos_gpu.py against the class compiled in place.
#include "../../include/sdhip.h"
#include "common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace sdhip
{
    constexpr int AOS_MAXP = 32; // packet headers recorded per frame (a frame with more is scanned on the host)
    struct AosScan
    {
        unsigned short fhp;
        unsigned short npk;          // chain entries recorded (0xFFFF: more than AOS_MAXP)
        unsigned short partial_pos;  // a header that does not fit (pos + 6 >= size, pos < size): its position, else 0xFFFF
        unsigned char first6[6];     // the zone's first six bytes
        unsigned short pos[AOS_MAXP];
        unsigned char hdr[AOS_MAXP][6];
    };

    __global__ __launch_bounds__(256) void k_aos_vcdu(const unsigned char *cadus, int cadu_bytes, int nframes, sdhip_vcdu *out)
    {
        const int f = (int)(blockIdx.x * 256 + threadIdx.x);
        if (f >= nframes)
            return;
        const unsigned char *c = cadus + (size_t)f * cadu_bytes;
        sdhip_vcdu v;
        v.version = c[4] >> 6;
        v.spacecraft_id = (unsigned short)(((c[4] & 0x3F) << 2) | (c[5] >> 6));
        v.vcid = c[5] & 0x3F;
        v.vcdu_counter = ((unsigned)c[6] << 16) | ((unsigned)c[7] << 8) | c[8];
        v.replay_flag = c[9] >> 7;
        out[f] = v;
    }
    // frames whose VCID is `vcid`, in order: block-wide scan of the match flags, one atomic per block for the blocks' order (blocks are ordered by a
    // second pass over the block counts: two launches)
    __global__ __launch_bounds__(256) void k_aos_count(const unsigned char *cadus, int cadu_bytes, int nframes, int vcid, int *block_counts)
    {
        __shared__ int cnt;
        if (threadIdx.x == 0)
            cnt = 0;
        __syncthreads();
        const int f = (int)(blockIdx.x * 256 + threadIdx.x);
        if (f < nframes && (cadus[(size_t)f * cadu_bytes + 5] & 0x3F) == vcid)
            atomicAdd(&cnt, 1);
        __syncthreads();
        if (threadIdx.x == 0)
            block_counts[blockIdx.x] = cnt;
    }
    __global__ __launch_bounds__(256) void k_aos_select(const unsigned char *cadus, int cadu_bytes, int nframes, int vcid, const int *block_offsets, unsigned char *out, int *index_out)
    {
        __shared__ int flags[256];
        const int f = (int)(blockIdx.x * 256 + threadIdx.x);
        const int m = (f < nframes && (cadus[(size_t)f * cadu_bytes + 5] & 0x3F) == vcid) ? 1 : 0;
        flags[threadIdx.x] = m;
        __syncthreads();
        int rank = 0;
        for (int i = 0; i < (int)threadIdx.x; i++)
            rank += flags[i];
        if (m)
        {
            const int o = block_offsets[blockIdx.x] + rank;
            const unsigned char *s = cadus + (size_t)f * cadu_bytes;
            unsigned char *d = out + (size_t)o * cadu_bytes;
            for (int b = 0; b < cadu_bytes; b++)
                d[b] = s[b];
            if (index_out)
                index_out[o] = f;
        }
    }
    __global__ __launch_bounds__(128) void k_aos_scan(const unsigned char *cadus, int cadu_bytes, int nframes, int zone_off, int size, int sec_hdr_extends, AosScan *out)
    {
        const int f = (int)(blockIdx.x * 128 + threadIdx.x);
        if (f >= nframes)
            return;
        const unsigned char *c = cadus + (size_t)f * cadu_bytes;
        const unsigned char *D = c + zone_off + 2; // mpdu.data (mpdu.cpp:12-13)
        AosScan &s = out[f];
        const int fhp = ((c[zone_off] & 7) << 8) | c[zone_off + 1];
        s.fhp = (unsigned short)fhp;
        for (int i = 0; i < 6; i++)
            s.first6[i] = D[i];
        int npk = 0, partial = 0xFFFF;
        if (fhp < 2047 && fhp < size)
        {
            int pos = fhp;
            while (pos < size)
            {
                if (pos + 6 < size)
                {
                    if (npk == AOS_MAXP)
                    {
                        npk = 0xFFFF;
                        break;
                    }
                    s.pos[npk] = (unsigned short)pos;
                    for (int i = 0; i < 6; i++)
                        s.hdr[npk][i] = D[pos + i];
                    const int plen = ((D[pos + 4] << 8) | D[pos + 5]) + 1 + ((sec_hdr_extends && ((D[pos] >> 3) & 1)) ? 8 : 0);
                    npk++;
                    pos += plen + 6;
                }
                else
                {
                    partial = pos;
                    break;
                }
            }
        }
        s.npk = (unsigned short)npk;
        s.partial_pos = (unsigned short)partial;
    }
    struct AosCopy
    {
        unsigned frame;    // index into the frames of this call
        unsigned short src; // offset in the zone
        unsigned short len;
        unsigned long long dst; // offset in the payload pool
    };
    __global__ __launch_bounds__(64) void k_aos_gather(const unsigned char *cadus, int cadu_bytes, int zone_off, const AosCopy *cmds, int ncmd, unsigned char *pool)
    {
        const int k = (int)blockIdx.x;
        if (k >= ncmd)
            return;
        const AosCopy c = cmds[k];
        const unsigned char *s = cadus + (size_t)c.frame * cadu_bytes + zone_off + 2 + c.src;
        unsigned char *d = pool + c.dst;
        for (int b = (int)threadIdx.x; b < (int)c.len; b += 64)
            d[b] = s[b];
    }

    // ---- the Demuxer's state machine on the scan records (demuxer.cpp:67-201). A packet under construction lives in `cur`; its payload so far is a list of
    // byte ranges (of this call's frames, or of bytes carried over from earlier calls: those are kept in `carry` on the host -- a packet spans a call
    // boundary at most once per call).
    struct AosDemux
    {
        int device, size, zone_off, sec_ext;
        // Demuxer state
        bool working = false, in_header = false;
        bool broken = false; // a capacity error left the state machine ahead of what the caller received: the handle refuses further work
        int remaining = 0, payload_len = 0, total_len = 0, in_header_n = 0;
        unsigned char header_buf[6] = {0, 0, 0, 0, 0, 0};
        unsigned char cur_hdr[6] = {0, 0, 0, 0, 0, 0};
        std::vector<unsigned char> carry; // payload bytes of the packet under construction that came from earlier calls
        // per call
        struct Range
        {
            unsigned frame;
            unsigned short src, len;
        };
        std::vector<Range> cur_ranges;
        std::vector<sdhip_aos_packet> packets;
        std::vector<AosCopy> cmds;
        std::vector<unsigned char> host_bytes; // carried bytes go out through a host-side pool segment
        std::vector<std::pair<unsigned long long, unsigned>> host_segs; // (dst, length) of carried bytes, in host_bytes order
        unsigned long long pool_used = 0;
        DevBuf<AosScan> d_scan;
        DevBuf<AosCopy> d_cmds;
        std::vector<AosScan> h_scan;

        void read_packet(const unsigned char *h)
        { // Demuxer::readPacket, demuxer.cpp:24-31
            working = true;
            memcpy(cur_hdr, h, 6);
            const int plen = (h[4] << 8) | h[5];
            payload_len = plen + 1 + ((sec_ext && ((h[0] >> 3) & 1)) ? 8 : 0);
            total_len = payload_len + 6;
            remaining = payload_len;
        }
        void push_payload(unsigned frame, int src, int len)
        { // Demuxer::pushPayload, :44-50 (a non-positive length copies nothing but still moves `remaining`)
            if (len > 0)
                cur_ranges.push_back(Range{frame, (unsigned short)src, (unsigned short)len});
            remaining -= len;
        }
        void clear_cur()
        {
            cur_ranges.clear();
            carry.clear();
            payload_len = 0;
            remaining = 0;
            working = false;
        }
        void push_packet(unsigned frame)
        { // Demuxer::pushPacket, :34-41: the packet leaves with whatever payload it has
            sdhip_aos_packet p;
            memset(&p, 0, sizeof(p));
            memcpy(p.header, cur_hdr, 6);
            p.version = cur_hdr[0] >> 5;
            p.type = (cur_hdr[0] >> 4) & 1;
            p.secondary_header_flag = (cur_hdr[0] >> 3) & 1;
            p.apid = (unsigned short)(((cur_hdr[0] & 7) << 8) | cur_hdr[1]);
            p.sequence_flag = cur_hdr[2] >> 6;
            p.packet_sequence_count = (unsigned short)(((cur_hdr[2] & 0x3F) << 8) | cur_hdr[3]);
            p.packet_length = (unsigned short)((cur_hdr[4] << 8) | cur_hdr[5]);
            p.frame = frame;
            p.payload_offset = pool_used;
            unsigned long long n = 0;
            if (!carry.empty())
            {
                host_segs.push_back({pool_used, (unsigned)carry.size()});
                host_bytes.insert(host_bytes.end(), carry.begin(), carry.end());
                n += carry.size();
            }
            for (auto &r : cur_ranges)
            {
                cmds.push_back(AosCopy{r.frame, r.src, r.len, pool_used + n});
                n += r.len;
            }
            p.payload_size = (unsigned)n;
            pool_used += n;
            packets.push_back(p);
            clear_cur();
        }
        void abort_packet() { clear_cur(); }

        // one frame (Demuxer::work)
        void frame(unsigned f, const AosScan &s, const unsigned char *zone_host /* only for frames the scan overflowed on */)
        {
            const int fhp = s.fhp;
            if (fhp < 2047 && fhp >= size)
                return;
            int offset = 0;
            if (in_header)
            {
                in_header = false;
                memcpy(header_buf + in_header_n, s.first6, (size_t)(6 - in_header_n));
                offset = 6 - in_header_n;
                in_header_n = 6;
                read_packet(header_buf);
            }
            if (remaining > 0 && working)
            {
                if (fhp < 2047)
                {
                    const int to_write = (remaining + offset) > fhp + 1 ? (fhp + 1) - offset : remaining;
                    push_payload(f, offset, to_write);
                    remaining = 0;
                }
                else
                {
                    const int to_write = (remaining + offset) > size - offset ? size - offset : remaining;
                    push_payload(f, offset, to_write);
                }
            }
            if (remaining == 0 && working)
                push_packet(f);
            if (fhp >= 2047)
                return;
            // the chain of headers this frame starts: from the scan record, or (more than AOS_MAXP of them) read off the zone on the host
            auto hdr_at = [&](int k, int pos, unsigned char *h6) {
                if (zone_host)
                    memcpy(h6, zone_host + pos, 6);
                else
                    memcpy(h6, s.hdr[k], 6);
            };
            if (fhp + 6 < size)
            {
                unsigned char h6[6];
                hdr_at(0, fhp, h6);
                read_packet(h6);
                const bool has_second = size > fhp + total_len;
                if (has_second)
                {
                    if (fhp + total_len < size)
                    {
                        push_payload(f, fhp + 6, payload_len);
                        push_packet(f);
                    }
                    else
                        abort_packet();
                    int next = fhp + total_len, k = 1;
                    while (next < size)
                    {
                        if (next + 6 < size)
                        {
                            hdr_at(k, next, h6);
                            k++;
                            read_packet(h6);
                            const int to_write = remaining > (size - (next + 6)) ? (size - (next + 6)) : remaining;
                            push_payload(f, next + 6, to_write);
                        }
                        else if (next < size)
                        {
                            in_header = true;
                            in_header_n = size - next;
                            if (zone_host)
                                memcpy(header_buf, zone_host + next, (size_t)in_header_n);
                            else
                                memcpy(header_buf, partial_bytes, (size_t)in_header_n);
                            break;
                        }
                        if (remaining == 0 && working)
                            push_packet(f);
                        next = next + total_len;
                    }
                }
                else if (working)
                {
                    const int to_write = remaining > (size - (fhp + 6)) ? (size - (fhp + 6)) : remaining;
                    push_payload(f, fhp + 6, to_write);
                }
            }
            else if (fhp < size)
            {
                in_header = true;
                in_header_n = size - fhp;
                if (zone_host)
                    memcpy(header_buf, zone_host + fhp, (size_t)in_header_n);
                else
                    memcpy(header_buf, partial_bytes, (size_t)in_header_n);
            }
        }
        unsigned char partial_bytes[6] = {0, 0, 0, 0, 0, 0}; // the bytes of a header that does not fit, of the frame being walked
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
    int sdhip_aos_parse_vcdu_dev(int device, const uint8_t *d_cadus, int cadu_bytes, int nframes, sdhip_vcdu *d_out)
    {
        SD_GUARD_BEGIN
        if (cadu_bytes < 10 || nframes < 0)
            throw HipError("aos: a CADU has at least the sync marker and the VCDU primary header");
        SD_HIP(hipSetDevice(device));
        if (nframes)
        {
            ProfScope _ps("k_aos_vcdu", nullptr);
            hipLaunchKernelGGL(k_aos_vcdu, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, nullptr, d_cadus, cadu_bytes, nframes, d_out);
        }
        SD_HIP(hipDeviceSynchronize());
        return 0;
        SD_GUARD_END(-1)
    }
    int64_t sdhip_aos_select_vcid_dev(int device, const uint8_t *d_cadus, int cadu_bytes, int nframes, int vcid, uint8_t *d_out, size_t cap_frames, int *d_index_out)
    {
        SD_GUARD_BEGIN
        if (cadu_bytes < 10 || nframes < 0)
            throw HipError("aos: a CADU has at least the sync marker and the VCDU primary header");
        if (nframes == 0)
            return 0;
        SD_HIP(hipSetDevice(device));
        const int nb = (nframes + 255) / 256;
        DevBuf<int> d_cnt;
        d_cnt.reserve((size_t)nb);
        hipLaunchKernelGGL(k_aos_count, dim3((unsigned)nb), dim3(256), 0, nullptr, d_cadus, cadu_bytes, nframes, vcid, d_cnt.p);
        std::vector<int> cnt((size_t)nb);
        SD_HIP(hipMemcpy(cnt.data(), d_cnt.p, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost));
        long long total = 0;
        for (int b = 0; b < nb; b++)
        {
            const int c = cnt[b];
            cnt[b] = (int)total;
            total += c;
        }
        if ((size_t)total > cap_frames)
            throw HipError("aos select: output buffer too small");
        SD_HIP(hipMemcpy(d_cnt.p, cnt.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice));
        {
            ProfScope _ps("k_aos_select", nullptr);
            hipLaunchKernelGGL(k_aos_select, dim3((unsigned)nb), dim3(256), 0, nullptr, d_cadus, cadu_bytes, nframes, vcid, d_cnt.p, d_out, d_index_out);
        }
        SD_HIP(hipDeviceSynchronize());
        return (int64_t)total;
        SD_GUARD_END(-1)
    }
    void *sdhip_aos_demux_create(int device, int mpdu_data_size, int has_insert_zone, int insert_zone_size, int secondary_header_extends_pkt)
    {
        SD_GUARD_BEGIN
        if (mpdu_data_size < 7 || mpdu_data_size > 2046 || insert_zone_size < 0)
            throw HipError("aos demux: mpdu_data_size out of range");
        AosDemux *d = new AosDemux;
        d->device = device;
        d->size = mpdu_data_size;
        d->zone_off = has_insert_zone ? 10 + insert_zone_size : 10; // mpdu.cpp:12
        d->sec_ext = secondary_header_extends_pkt ? 1 : 0;
        return d;
        SD_GUARD_END(nullptr)
    }
    void sdhip_aos_demux_destroy(void *h) { delete (AosDemux *)h; }
    int64_t sdhip_aos_demux_work_dev(void *h, const uint8_t *d_cadus, int cadu_bytes, int nframes, sdhip_aos_packet *packets_out, size_t cap_packets, uint8_t *d_payload, size_t cap_payload,
                                     uint64_t *payload_bytes_out)
    {
        SD_GUARD_BEGIN
        AosDemux &d = *(AosDemux *)h;
        if (d.broken) // (a capacity error comes after the state machine has consumed the call's frames -- a retry would feed them into an advanced state)
            throw HipError("aos demux: an earlier call ran out of output capacity in mid-stream; the handle's state is behind the frames it has seen -- create a new one");
        if (cadu_bytes < d.zone_off + 2 + d.size)
            throw HipError("aos demux: the CADU is shorter than its M_PDU zone");
        d.packets.clear();
        d.cmds.clear();
        d.host_bytes.clear();
        d.host_segs.clear();
        d.pool_used = 0;
        if (payload_bytes_out)
            *payload_bytes_out = 0;
        if (nframes <= 0)
            return 0;
        SD_HIP(hipSetDevice(d.device));
        d.d_scan.reserve((size_t)nframes);
        {
            ProfScope _ps("k_aos_scan", nullptr);
            hipLaunchKernelGGL(k_aos_scan, dim3((unsigned)((nframes + 127) / 128)), dim3(128), 0, nullptr, d_cadus, cadu_bytes, nframes, d.zone_off, d.size, d.sec_ext, d.d_scan.p);
        }
        d.h_scan.resize((size_t)nframes);
        SD_HIP(hipMemcpy(d.h_scan.data(), d.d_scan.p, (size_t)nframes * sizeof(AosScan), hipMemcpyDeviceToHost));
        // a packet under construction from the previous call: its ranges pointed into that call's frames -- they were turned into carried bytes there
        std::vector<unsigned char> zone;
        for (int f = 0; f < nframes; f++)
        {
            const AosScan &s = d.h_scan[(size_t)f];
            const unsigned char *zh = nullptr;
            if (s.npk == 0xFFFF)
            { // more packet headers than the record holds: read this frame's zone
                zone.resize((size_t)d.size);
                SD_HIP(hipMemcpy(zone.data(), d_cadus + (size_t)f * cadu_bytes + d.zone_off + 2, (size_t)d.size, hipMemcpyDeviceToHost));
                zh = zone.data();
            }
            else if (s.partial_pos != 0xFFFF)
            { // the bytes of the header that does not fit: the record's last six zone bytes are not kept, fetch them (rare: a header on a zone's last 6 bytes)
                const int n = d.size - s.partial_pos;
                SD_HIP(hipMemcpy(d.partial_bytes, d_cadus + (size_t)f * cadu_bytes + d.zone_off + 2 + s.partial_pos, (size_t)n, hipMemcpyDeviceToHost));
            }
            d.frame((unsigned)f, s, zh);
        }
        if (d.packets.size() > cap_packets || d.pool_used > cap_payload)
        {
            d.broken = true;
            throw HipError(d.packets.size() > cap_packets ? "aos demux: packet table too small (the handle must be created anew)" : "aos demux: payload pool too small (the handle must be created anew)");
        }
        if (!d.cmds.empty())
        {
            d.d_cmds.reserve(d.cmds.size());
            SD_HIP(hipMemcpy(d.d_cmds.p, d.cmds.data(), d.cmds.size() * sizeof(AosCopy), hipMemcpyHostToDevice));
            ProfScope _ps("k_aos_gather", nullptr);
            hipLaunchKernelGGL(k_aos_gather, dim3((unsigned)d.cmds.size()), dim3(64), 0, nullptr, d_cadus, cadu_bytes, d.zone_off, d.d_cmds.p, (int)d.cmds.size(), d_payload);
        }
        size_t hb = 0;
        for (auto &seg : d.host_segs)
        {
            SD_HIP(hipMemcpy(d_payload + seg.first, d.host_bytes.data() + hb, seg.second, hipMemcpyHostToDevice));
            hb += seg.second;
        }
        // the packet still under construction keeps its bytes across the call: fetch the ranges it has so far (at most one packet's worth)
        if (d.working && !d.cur_ranges.empty())
        {
            for (auto &r : d.cur_ranges)
            {
                const size_t o = d.carry.size();
                d.carry.resize(o + r.len);
                SD_HIP(hipMemcpy(d.carry.data() + o, d_cadus + (size_t)r.frame * cadu_bytes + d.zone_off + 2 + r.src, r.len, hipMemcpyDeviceToHost));
            }
            d.cur_ranges.clear();
        }
        SD_HIP(hipDeviceSynchronize());
        memcpy(packets_out, d.packets.data(), d.packets.size() * sizeof(sdhip_aos_packet));
        if (payload_bytes_out)
            *payload_bytes_out = d.pool_used;
        return (int64_t)d.packets.size();
        SD_GUARD_END(-1)
    }
}
