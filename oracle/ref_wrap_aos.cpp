// oracle/ref_wrap_aos.cpp -- TEST INFRASTRUCTURE ONLY: the reference's CCSDS AOS helpers compiled in place (src-core/common/ccsds/ccsds.cpp,
// ccsds_aos/{vcdu,mpdu,demuxer}.cpp): parseVCDU per frame, and one Demuxer instance fed frame by frame the way the instrument decoders feed it.
#include "common/ccsds/ccsds_aos/demuxer.h"
#include "common/ccsds/ccsds_aos/vcdu.h"
#include <cstring>

extern "C"
{
    // out5[f] = {version, spacecraft_id, vcid, vcdu_counter, replay_flag}
    void sdref_aos_vcdu(const uint8_t *cadus, int cadu_bytes, int nframes, unsigned *out5)
    {
        for (int f = 0; f < nframes; f++)
        {
            ccsds::ccsds_aos::VCDU v = ccsds::ccsds_aos::parseVCDU((uint8_t *)cadus + (size_t)f * cadu_bytes);
            out5[5 * f + 0] = v.version;
            out5[5 * f + 1] = v.spacecraft_id;
            out5[5 * f + 2] = v.vcid;
            out5[5 * f + 3] = v.vcdu_counter;
            out5[5 * f + 4] = v.replay_flag;
        }
    }
    // Demuxer::work over nframes frames. Per packet: hdr_out 6 bytes, meta_out {frame, payload size, apid, sequence count, packet_length, sequence_flag}; payload bytes appended to pool.
    long long sdref_aos_demux(int mpdu_data_size, int has_insert_zone, int insert_zone_size, int sec_ext, const uint8_t *cadus, int cadu_bytes, int nframes, uint8_t *hdr_out,
                              unsigned *meta_out, long long cap_packets, uint8_t *pool, long long cap_pool, long long *pool_used)
    {
        ccsds::ccsds_aos::Demuxer d(mpdu_data_size, has_insert_zone, insert_zone_size, sec_ext);
        long long np = 0, used = 0;
        for (int f = 0; f < nframes; f++)
        {
            std::vector<ccsds::CCSDSPacket> pk = d.work((uint8_t *)cadus + (size_t)f * cadu_bytes);
            for (auto &p : pk)
            {
                if (np >= cap_packets || used + (long long)p.payload.size() > cap_pool)
                    return -2;
                memcpy(hdr_out + 6 * np, p.header.raw, 6);
                meta_out[6 * np + 0] = (unsigned)f;
                meta_out[6 * np + 1] = (unsigned)p.payload.size();
                meta_out[6 * np + 2] = p.header.apid;
                meta_out[6 * np + 3] = p.header.packet_sequence_count;
                meta_out[6 * np + 4] = p.header.packet_length;
                meta_out[6 * np + 5] = p.header.sequence_flag;
                if (!p.payload.empty())
                    memcpy(pool + used, p.payload.data(), p.payload.size());
                used += (long long)p.payload.size();
                np++;
            }
        }
        *pool_used = used;
        return np;
    }
}
