// shard.hip -- stream-parallel sharding of ONE recording over several devices / ranks (north_star: "independent baseband chunks shard
// stream-parallel across the 8 GPUs of one node, overlap regions stitched on host"), behind the C ABI since round 4 (include/sdhip.h, sdhip_shard_*):
// host logic only, no kernel. The reference's topology for ONE stream is a thread per module joined by FIFOs (src-core/pipeline/pipeline_run.cpp:72-104);
// sharding in time is what a data-parallel machine adds, and it has to end in what the single stream produces:
//   plan     contiguous per-chunk sample ranges; a chunk reads `overlap` samples in front of its own range (its loops and decoders lock in there)
//   align    where a chunk's soft-symbol stream CONTINUES its predecessor's: both demodulated the overlap's samples, so the predecessor's last symbols
//            appear in the chunk's stream (at some lag, turned by a constellation symmetry -- the two carrier loops locked independently); from that lag
//            every chunk knows the GLOBAL index of its symbols, and starts its decoder on the single stream's own Viterbi block grid (a multiple of the
//            decoder's buffer): block boundaries are where a block decoder's output depends on the cut (cc_decoder.cpp:239-273 traces back with 6 steps of
//            look-ahead), so only on that grid do N chunks decode the bits one stream decodes
//   stitch   frames to drop at the head of every chunk's CADU list: the frames its predecessor decoded too (compared whole -- sync marker and RS parity
//            included -- when the decoders ran on the common grid)
#include "../../include/sdhip.h"
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <string>
#include <vector>

namespace sdhip
{
    static inline int hard_bit(int8_t v) { return v < 0 ? 1 : 0; }
    // (I, Q) turned by t quarter turns counter-clockwise, as signs: what the decoders' rotate_soft does to the soft symbols (rotation.cpp:4-63)
    static inline void turn_iq(int t, int i, int q, int &oi, int &oq)
    {
        switch (t & 3)
        {
        case 0:
            oi = i, oq = q;
            break;
        case 1:
            oi = -q, oq = i;
            break;
        case 2:
            oi = -i, oq = -q;
            break;
        default:
            oi = q, oq = -i;
            break;
        }
    }
    static bool frames_equal(const uint8_t *a, const uint8_t *b, int frame_bytes, bool whole)
    {
        const int skip = (!whole && frame_bytes > 8) ? 4 : 0; // the sync marker is not RS protected
        return memcmp(a + skip, b + skip, (size_t)(frame_bytes - skip)) == 0;
    }
    // leading frames of `head` that repeat the end of `tail_prev`: the largest m with head[:m] == tail_prev[-m:]; if there is none, the first frames of
    // head that occur anywhere in tail_prev (the chunk's re-lock began in the middle of the overlap)
    static size_t overlap_drop(const std::vector<const uint8_t *> &tail_prev, const std::vector<const uint8_t *> &head, int fb, bool whole)
    {
        if (tail_prev.empty() || head.empty())
            return 0;
        for (size_t p = 0; p < tail_prev.size(); p++)
        { // earliest position = largest overlap first
            if (!frames_equal(tail_prev[p], head[0], fb, whole))
                continue;
            const size_t m = tail_prev.size() - p;
            if (m > head.size())
                continue;
            bool all = true;
            for (size_t k = 0; k < m && all; k++)
                all = frames_equal(tail_prev[p + k], head[k], fb, whole);
            if (all)
                return m;
        }
        const int skip = (!whole && fb > 8) ? 4 : 0;
        std::set<std::string> seen;
        for (auto t : tail_prev)
            seen.insert(std::string((const char *)t + skip, (size_t)(fb - skip)));
        size_t j = 0;
        while (j < head.size() && seen.count(std::string((const char *)head[j] + skip, (size_t)(fb - skip))))
            j++;
        return j;
    }
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
    int sdhip_shard_lockin(const sdhip_demod_cfg *d, const sdhip_fec_cfg *f, uint64_t *out3)
    {
        // The lock-in times of the stages from the loop constants the modules are configured with. out3[0] = samples until a cold-started demodulator
        // produces the single stream's symbols: AGC 8 / agc_rate samples, Costas 16 / pll_bw samples, M&M 40 / clock_gain_mu symbols. out3[1] = soft BYTES a
        // cold-started decoder needs in front of the first frame that counts: viterbi_outsync_after + 2 Viterbi blocks (the lock search runs on the first block,
        // viterbi_1_2.cpp:52-92 -- when it runs while the loops still settle it can lock on a wrong phase with a BER just under the threshold, which the
        // decoder gives up after outsync_after bad blocks, :104-113), for MetOp 10 more (the module's no-sync watchdog,
        // module_metop_ahrpt_decoder.cpp:58-66), one more for the start on the block grid, and four CADUs (the deframer needs consecutive markers before it
        // reports SYNCED, bpsk_ccsds_deframer.cpp:47-107; one more frame straddles the boundary). out3[2] = the decoder's block in soft bytes
        // (max(cadu_size, 8192), 16384 for MetOp: what the modules read per buffer).
        SD_GUARD_BEGIN
        if (!d || !f || !out3 || !(d->samplerate > 0) || !(d->symbolrate > 0) || !(d->pll_bw > 0))
            throw HipError("shard lockin: samplerate, symbolrate and pll_bw must be set");
        const double sps = d->samplerate / d->symbolrate;
        const int q = d->constellation == SDHIP_BPSK ? 1 : 2;
        const bool metop = f->decoder == SDHIP_DEC_METOP_AHRPT;
        // the FengYun-3 modules (module_fengyun_ahrpt_decoder.cpp:10,63: BUFFER_SIZE * 2 = 16384 soft bytes per read, a Viterbi per rail over 8192 of them; the MPT
        // module likewise): from cold, ten counted reads until `shift` may toggle (:82-93), the Viterbis' own search and give-up (outsync_after + 2), ten NOSYNC reads
        // of the deframer until `invert_branches` toggles (:105-114), one more read for the grid
        const bool fy = f->decoder == SDHIP_DEC_FENGYUN_AHRPT || f->decoder == SDHIP_DEC_FENGYUN_MPT;
        const int cadu_bits = (metop || fy) ? 8192 : (f->cadu_size > 0 ? f->cadu_size : 8192);
        static const double rates[5] = {0.5, 2.0 / 3.0, 0.75, 5.0 / 6.0, 7.0 / 8.0};
        const double conv_rate = metop ? 0.75 : (f->decoder == SDHIP_DEC_FENGYUN_AHRPT ? 0.75 : (f->decoder == SDHIP_DEC_FENGYUN_MPT ? 0.5 : rates[f->conv_rate >= 0 && f->conv_rate <= 4 ? f->conv_rate : 0]));
        const uint64_t block_bytes = (metop || fy) ? 16384u : (uint64_t)std::max(cadu_bits, 8192);
        const double cadu_bytes = cadu_bits / conv_rate; // soft bytes per CADU: cadu_bits / rate symbols-worth of soft bits, one byte each
        const double gmu = d->clock_gain_mu > 0 ? d->clock_gain_mu : 8.7e-3;
        const int outsync = f->viterbi_outsync_after > 0 ? f->viterbi_outsync_after : (metop ? 10 : 20);
        const int relock_blocks = outsync + 2 + (metop ? 10 : 0) + (fy ? 20 : 0) + 1;
        const double agc = d->agc_rate > 0 ? d->agc_rate : 1e-2;
        out3[0] = (uint64_t)(8.0 / agc + 16.0 / d->pll_bw + 40.0 / gmu * sps + 0.5);
        out3[1] = (uint64_t)(relock_blocks * (double)block_bytes + 4 * cadu_bytes + 0.5);
        out3[2] = block_bytes;
        (void)q;
        return 0;
        SD_GUARD_END(-1)
    }
    uint64_t sdhip_shard_overlap(const sdhip_demod_cfg *d, const sdhip_fec_cfg *f)
    {
        uint64_t p[3];
        if (sdhip_shard_lockin(d, f, p) != 0)
            return 0;
        // decoder lock-in in samples: its soft bytes are q per symbol, a symbol is sps samples
        const double sps = d->samplerate / d->symbolrate;
        const int q = d->constellation == SDHIP_BPSK ? 1 : 2;
        const double n = (double)p[0] + (double)p[1] / q * sps;
        return (uint64_t)((long long)(n + 7) / 8 * 8);
    }
    int sdhip_shard_plan(uint64_t n_samples, int world, uint64_t overlap, int align, sdhip_shard_range *out)
    {
        SD_GUARD_BEGIN
        if (world < 1 || !out)
            throw HipError("shard plan: world must be >= 1");
        if (align < 1)
            align = 8;
        std::vector<uint64_t> edges(world + 1);
        for (int r = 0; r < world; r++)
            edges[r] = (uint64_t)((unsigned __int128)n_samples * r / world) / align * align;
        edges[world] = n_samples;
        for (int r = 0; r < world; r++)
        {
            out[r].own_start = edges[r];
            out[r].stop = edges[r + 1];
            out[r].read_start = edges[r] > overlap ? edges[r] - overlap : 0;
        }
        return 0;
        SD_GUARD_END(-1)
    }
    int sdhip_shard_align(const int8_t *prev_tail, size_t n_prev, const int8_t *head, size_t n_head, int q, int64_t expect, int64_t radius, int64_t *lag_symbols, int *turn,
                          float *agreement)
    {
        SD_GUARD_BEGIN
        if ((q != 1 && q != 2) || !prev_tail || !head || !lag_symbols)
            throw HipError("shard align: q must be 1 (BPSK) or 2, buffers must be there");
        const int64_t T = (int64_t)(n_prev / q), H = (int64_t)(n_head / q);
        if (T < 64 || H < T)
            throw HipError("shard align: the predecessor's tail must hold at least 64 symbols and the head at least as many");
        // pattern = the predecessor's last T symbols; candidate p = index in head of the symbol that FOLLOWS the pattern (the continuation point):
        // head[p - T .. p) against the pattern
        int64_t lo = std::max<int64_t>(T, expect - radius), hi = std::min<int64_t>(H, expect + radius);
        if (radius <= 0)
        {
            lo = T;
            hi = H;
        }
        int64_t best_p = -1;
        int best_t = 0;
        long best = -1;
        const int nturn = q == 1 ? 2 : 4;
        for (int64_t p = lo; p <= hi; p++)
        {
            const int8_t *h = head + (size_t)(p - T) * q;
            long agree[4] = {0, 0, 0, 0};
            if (q == 1)
            {
                long a = 0;
                for (int64_t k = 0; k < T; k++)
                    a += hard_bit(h[k]) == hard_bit(prev_tail[k]);
                agree[0] = a;
                agree[1] = T - a; // half a turn
            }
            else
            {
                for (int64_t k = 0; k < T; k++)
                {
                    const int hi_ = h[2 * k] < 0 ? -1 : 1, hq = h[2 * k + 1] < 0 ? -1 : 1;
                    const int ti = prev_tail[2 * k] < 0 ? -1 : 1, tq = prev_tail[2 * k + 1] < 0 ? -1 : 1;
                    for (int t = 0; t < 4; t++)
                    {
                        int oi, oq;
                        turn_iq(t, hi_, hq, oi, oq);
                        agree[t] += (oi == ti) + (oq == tq);
                    }
                }
            }
            for (int t = 0; t < nturn; t++)
                if (agree[t] > best)
                {
                    best = agree[t];
                    best_p = p;
                    best_t = q == 1 ? 2 * t : t;
                }
        }
        const double frac = best < 0 ? 0.0 : (double)best / (double)(T * q);
        if (agreement)
            *agreement = (float)frac;
        *lag_symbols = best_p;
        if (turn)
            *turn = best_t;
        return frac >= 0.9 ? 0 : 1; // 1: no continuation found (the two chains did not both lock on the overlap)
        SD_GUARD_END(-1)
    }
    int sdhip_shard_stitch(const uint8_t *const *heads, const size_t *n_heads, const uint8_t *const *tails, const size_t *n_tails, const uint64_t *counts, int world, int frame_bytes,
                           size_t edge, int whole_frames, uint64_t *drops_out)
    {
        SD_GUARD_BEGIN
        if (world < 1 || frame_bytes < 1 || !drops_out)
            throw HipError("shard stitch: bad arguments");
        const bool whole = whole_frames != 0;
        std::vector<std::vector<uint8_t>> store; // the running tail of the stitched stream owns copies
        std::vector<const uint8_t *> run;
        auto set_run = [&](const std::vector<const uint8_t *> &fr) {
            std::vector<std::vector<uint8_t>> ns;
            for (auto p : fr)
                ns.emplace_back(p, p + frame_bytes);
            store.swap(ns);
            run.clear();
            for (auto &v : store)
                run.push_back(v.data());
        };
        for (int r = 0; r < world; r++)
        {
            drops_out[r] = 0;
            const size_t c = (size_t)counts[r];
            if (c == 0)
                continue;
            std::vector<const uint8_t *> h, t;
            for (size_t k = 0; k < n_heads[r]; k++)
                h.push_back(heads[r] + k * (size_t)frame_bytes);
            for (size_t k = 0; k < n_tails[r]; k++)
                t.push_back(tails[r] + k * (size_t)frame_bytes);
            if (!run.empty())
            {
                const size_t d = overlap_drop(run, h, frame_bytes, whole);
                drops_out[r] = d;
                if (h.size() >= edge && c > h.size())
                {
                    // every boundary frame of this chunk repeats the predecessor, or the predecessor's oldest kept frame shows up inside this chunk's
                    // head: the overlap reaches beyond the `edge` frames that were exchanged
                    bool oldest_inside = false;
                    if (d == 0)
                        for (auto p : h)
                            oldest_inside |= frames_equal(p, run[0], frame_bytes, whole);
                    if (d >= h.size() || oldest_inside)
                        throw HipError("shard stitch: chunk " + std::to_string(r) + ": the overlap with its predecessor exceeds the " + std::to_string(edge) +
                                       " boundary frames exchanged");
                }
            }
            const size_t kept = c - (size_t)drops_out[r];
            std::vector<const uint8_t *> kt; // the kept part of this chunk's tail
            if (kept >= t.size())
                kt = t;
            else
                kt.assign(t.end() - (long)kept, t.end());
            std::vector<const uint8_t *> nr;
            if (kept >= edge || run.empty())
                nr = kt;
            else
            {
                nr = run;
                nr.insert(nr.end(), kt.begin(), kt.end());
            }
            if (nr.size() > edge)
                nr.erase(nr.begin(), nr.end() - (long)edge);
            set_run(nr);
        }
        return 0;
        SD_GUARD_END(-1)
    }
}
