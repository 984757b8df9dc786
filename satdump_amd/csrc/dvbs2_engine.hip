// dvbs2_engine.hip -- the DVB-S2 demodulator as ONE handle (BASELINE.json configs[4]): what satdump::pipeline::dvb::DVBS2DemodModule
// (plugins/dvb_support/dvbs2/module_dvbs2_demod.{h,cpp}) builds in init() and runs in process() / process_s2() / process_s2_bch(), baseband samples in,
// BBFRAME bytes out:
//   BaseDemodModule's stages + RRC filter + M&M clock recovery   (module_dvbs2_demod.cpp:98-102)     -> the psk_demod engine without its Costas loop
//   FreqShiftBlock fed by the PLL's frequency (freq_prop_factor)  (:105, :204-206)                     -> k_s2_rotate, deterministic hand-over (below)
//   S2PLSyncBlock                                                 (:108-109, dvbs2_pl_sync.cpp)        -> sdhip_s2_pl_sync_dev
//   S2PLLBlock                                                    (:112-118, dvbs2_pll.cpp)            -> S2Pll: frame-parallel lanes, or the serial lane (exact)
//   S2BBToSoft                                                    (:121-126, dvbs2_bb_to_soft.cpp)     -> sdhip_s2_bb_to_soft_dev
//   process_s2: BBFrameLDPC::decode in groups of simd_type::SIZE frames, repack, BBFrameBCH::decode, BBFrameDescrambler (:239-293)
//                                                                                                      -> sdhip_ldpc_* / sdhip_s2_pack_dev / sdhip_bch_* / sdhip_bb_descramble_dev
// The handle owns the carry-over between calls: the PL synchroniser's ring (symbols not yet consumed), the PLL state, frames waiting for a full
// decoder group, the rotator's phase. Everything stays in HBM between the stages; the host sees BBFRAMEs and a handful of statistics.
//
// freq_prop_factor. The reference subtracts factor * PLL frequency from a rotator in front of the PL synchroniser once per frame that leaves
// S2BBToSoft -- from the module's thread, while the block threads run ahead by however many buffers their FIFOs hold: its output depends on thread
// timing. Here the hand-over happens at call boundaries, with the closed form of "once per frame, the PLL following at once": after a call that
// produced n frames the rotator takes over the fraction 1 - (1 - factor)^n of the PLL's frequency, and the PLL's frequency state is lowered by the
// same amount (the sum the symbols are turned by stays continuous -- the symbols waiting in the ring are turned on accordingly). Deterministic, stable for any n, and it converges to the same split: the
// rotator ends up carrying the offset, the PLL a residual near zero. factor = 0 is bit-for-bit the reference with that setting (exact mode).
#include "../../include/sdhip.h"
#include "common.h"
#include "dvbs2_stages.h"
#include "host_pipe.h"

#include <atomic>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace sdhip
{
    void demod_set_mm_windows(void *h, double tight, double tol); // demod_engine.hip
}
namespace sdhip
{
    // symbols turned by exp(j (phase0 + w i)): the rotator in front of the PL synchroniser
    __global__ __launch_bounds__(256) void k_s2_rotate(float2 *x, long long n, double phase0, double w)
    {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= n)
            return;
        double ph = phase0 + w * (double)i;
        ph -= 6.283185307179586476925 * rint(ph / 6.283185307179586476925);
        float s, c;
        sincosf((float)ph, &s, &c);
        const float2 v = x[i];
        x[i] = make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
    }
    // M2M4SNREstimator::update over a frame's slots (src-core/common/dsp/utils/snr_estimator.cpp:16-31): y <- alpha |x|^2 + beta y per symbol is an
    // exponential window; its value behind the frame = sum alpha beta^(N - 1 - i) |x_i|^k (+ beta^N times the value in front, 4e-10 of it for a normal
    // frame: dropped). A block per frame, double accumulation: a display statistic, float noise apart the reference's.
    __global__ __launch_bounds__(256) void k_s2_m2m4(const float2 *frames, int stride, int n, double alpha, double *out2)
    {
        __shared__ double s1[256], s2[256];
        const float2 *x = frames + (size_t)blockIdx.x * stride + 90;
        out2 += 2 * blockIdx.x;
        double a1 = 0.0, a2 = 0.0;
        const double lb = log1p(-alpha);
        for (int i = (int)threadIdx.x; i < n; i += 256)
        {
            const float2 v = x[i];
            const double p = (double)v.x * v.x + (double)v.y * v.y;
            const double w = alpha * exp(lb * (double)(n - 1 - i));
            a1 += w * p;
            a2 += w * p * p;
        }
        s1[threadIdx.x] = a1;
        s2[threadIdx.x] = a2;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1)
        {
            if ((int)threadIdx.x < st)
            {
                s1[threadIdx.x] += s1[threadIdx.x + st];
                s2[threadIdx.x] += s2[threadIdx.x + st];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
        {
            out2[0] = s1[0];
            out2[1] = s2[0];
        }
    }
    // BBFRAME bytes (the first kbch / 8 of every packed frame) gathered into the output rows
    __global__ __launch_bounds__(256) void k_s2_gather_bb(const unsigned char *pack, int pack_stride, int nbytes, int nframes, unsigned char *out)
    {
        const int f = (int)blockIdx.y, b = (int)(blockIdx.x * 256 + threadIdx.x);
        if (f < nframes && b < nbytes)
            out[(size_t)f * nbytes + b] = pack[(size_t)f * pack_stride + b];
    }

    struct Dvbs2Engine
    {
        sdhip_dvbs2_cfg cfg;
        S2Cfg mc;
        int device, raw, n_ldpc, k_ldpc, kbch, batch;
        void *front = nullptr, *ldpc = nullptr, *bch = nullptr;
        std::unique_ptr<S2Pll> pll;
        std::vector<int8_t> lut_bits;
        // carry-over
        DevBuf<float2> ring, ring2;
        size_t ring_len = 0, pl_spec = 64;
        DevBuf<int8_t> soft, soft2;
        size_t pend_frames = 0;
        double rot_phase = 0.0, current_freq = 0.0;
        // scratch
        DevBuf<int8_t> front_soft;
        DevBuf<float2> fr, pl;
        DevBuf<int> d_pls, d_tr, d_corr;
        DevBuf<unsigned char> d_pack, d_bb;
        DevBuf<double> d_m2m4;
        std::vector<double> h_m2m4;
        PinBuf<unsigned char> h_bb;
        std::vector<unsigned char> out_queue;
        size_t out_read = 0;
        sdhip_dvbs2_stats st{};
        std::mutex st_mu; // sdhip_dvbs2_demod_get_stats against the host path's worker thread: the getter sees the snapshot the last call left
        sdhip_dvbs2_stats st_pub{};
        int st_busy = 0; // calls in flight, under st_mu
        struct StScope
        {
            Dvbs2Engine &e;
            explicit StScope(Dvbs2Engine &en) : e(en)
            { // (see DemodEngine::StatsScope)
                std::lock_guard<std::mutex> lk(e.st_mu);
                if (e.st_busy++ == 0)
                    e.st_pub = e.st;
            }
            ~StScope()
            {
                std::lock_guard<std::mutex> lk(e.st_mu);
                if (--e.st_busy == 0)
                    e.st_pub = e.st;
            }
        };
        float peak_snr = 0.0f;
        static constexpr size_t HOST_BATCH = (size_t)8u << 20; // samples per shipped batch (~190 normal 8PSK frames at two samples per symbol)

        explicit Dvbs2Engine(const sdhip_dvbs2_cfg &c) : cfg(c)
        {
            mc = s2_cfg_of(c.modcod, c.shortframes ? 1 : 0); // get_dvbs2_cfg's messages
            device = c.front.device;
            if (!c.lut_bits || !c.lut_phase_error || c.lut_resolution < 2 || c.lut_resolution > 4096)
                throw HipError("dvbs2 demod: the demapper table (constellation_t::make_lut: bits and phase errors) must be handed over");
            if (c.ldpc_batch < 1 || c.ldpc_batch > 64)
                throw HipError("dvbs2 demod: ldpc_batch (the replaced build's simd_type::SIZE) out of range");
            if (!(c.freq_prop_factor >= 0.0f && c.freq_prop_factor < 1.0f))
                throw HipError("dvbs2 demod: freq_prop_factor must be in [0, 1)");
            if (c.front.exact && c.freq_prop_factor != 0.0f)
                throw HipError("dvbs2 demod: exact mode reproduces the reference bit for bit, which its thread-timed frequency feedback does not allow: freq_prop_factor must be 0");
            raw = s2_raw_frame_size(mc.slots, c.pilots ? 1 : 0);
            batch = c.ldpc_batch;
            SD_HIP(hipSetDevice(device));
            lut_bits.assign(c.lut_bits, c.lut_bits + (size_t)c.lut_resolution * c.lut_resolution * mc.bits);
            sdhip_demod_cfg f = c.front;
            f.constellation = SDHIP_QPSK;
            front = sdhip_dvbs2_front_create(&f);
            if (!front)
                throw HipError(std::string("dvbs2 demod: front end: ") + sdhip_last_error());
            // The clock recovery's hand-off windows, scaled to THIS loop: at the module's gain 1.7e-3 on a roll-off of 0.2 two trajectories of it hover ~1e-2 sample
            // apart for good (measured, visit G of round 5: 8 500 of 43 k boundaries outside 5e-3 behind 44 k samples of warm-up, dt 5e-3 .. 1.5e-2 behind 75 k), so
            // psk_demod's windows (2e-4 / 5e-3: its 1e-5 symbol contract) drove the adaptive warm-up to its cap -- 75 520 samples in front of 2 048-sample chunks, 97 %
            // of the lanes' work -- for nothing this module's consumers see: what is promised here is the decoders' output, and a symbol taken 2e-2 sample
            // off is 46 dB below the symbol. 2.5 / 5 interpolator arms instead (SDHIP_S2_MM_TIGHT_MILLI / _TOL_MILLI; 0 = psk_demod's).
            {
                const char *et = getenv("SDHIP_S2_MM_TIGHT_MILLI"), *eo = getenv("SDHIP_S2_MM_TOL_MILLI");
                const double tight = (et ? atof(et) : 20.0) * 1e-3, tol = (eo ? atof(eo) : 40.0) * 1e-3;
                if (!c.front.exact && tight > 0.0)
                    demod_set_mm_windows(front, tight, std::max(tol, tight));
            }
            sdhip_ldpc_cfg lc{c.shortframes ? 1 : 0, mc.rate, batch, device};
            ldpc = sdhip_ldpc_create(&lc);
            if (!ldpc)
                throw HipError(std::string("dvbs2 demod: ") + sdhip_last_error());
            sdhip_ldpc_info li;
            sdhip_ldpc_get_info(ldpc, &li);
            n_ldpc = li.code_len;
            k_ldpc = li.data_len;
            sdhip_bch_cfg bc{c.shortframes ? 1 : 0, mc.rate, device};
            bch = sdhip_bch_create(&bc);
            if (!bch)
                throw HipError(std::string("dvbs2 demod: ") + sdhip_last_error());
            int nb = 0;
            sdhip_bch_dims(bch, &kbch, &nb);
            pll.reset(new S2Pll(device, c.modcod, c.shortframes ? 1 : 0, c.pilots ? 1 : 0, c.front.pll_bw, c.lut_phase_error, c.lut_resolution, c.front.exact != 0));
            st.detected_modcod = -1;
        }
        ~Dvbs2Engine()
        {
            pipe.reset(); // the host path's worker thread first
            if (front)
                sdhip_demod_destroy(front);
            if (ldpc)
                sdhip_ldpc_destroy(ldpc);
            if (bch)
                sdhip_bch_destroy(bch);
        }
        int bbframe_bytes() const { return kbch / 8; }

        // clock-recovered symbols (device) -> BBFRAMEs appended to d_bb; returns the frames completed
        size_t feed_symbols(const float2 *d_syms, size_t nsym)
        {
            StScope _ss(*this);
            SD_HIP(hipSetDevice(device));
            if (nsym)
            {
                if (ring_len + nsym > ring.cap)
                { // grow, keeping the carried symbols
                    ring2.reserve(ring_len + nsym);
                    if (ring_len)
                        SD_HIP(hipMemcpy(ring2.p, ring.p, ring_len * sizeof(float2), hipMemcpyDeviceToDevice));
                    ring.swap(ring2);
                }
                SD_HIP(hipMemcpy(ring.p + ring_len, d_syms, nsym * sizeof(float2), hipMemcpyDeviceToDevice));
                if (cfg.freq_prop_factor != 0.0f)
                {
                    ProfScope _ps("k_s2_rotate", nullptr);
                    hipLaunchKernelGGL(k_s2_rotate, dim3((unsigned)((nsym + 255) / 256)), dim3(256), 0, nullptr, ring.p + ring_len, (long long)nsym, rot_phase, current_freq);
                    rot_phase = fmod(rot_phase + current_freq * (double)nsym, 6.283185307179586476925);
                }
                ring_len += nsym;
            }
            if (ring_len < (size_t)raw)
                return 0;
            // ---- PL synchroniser
            const size_t cap_frames = ring_len / raw + 1;
            fr.reserve(cap_frames * raw);
            pl.reserve(cap_frames * raw);
            size_t consumed = 0;
            const int64_t nf = s2_pl_sync_run(device, mc.slots, cfg.pilots ? 1 : 0, cfg.sof_thresold, reinterpret_cast<const float *>(ring.p), ring_len,
                                              reinterpret_cast<float *>(fr.p), raw, cap_frames, &consumed, nullptr, &pl_spec);
            if (consumed)
            {
                const size_t left = ring_len - consumed;
                ring2.reserve(std::max(left, (size_t)1));
                if (left)
                    SD_HIP(hipMemcpy(ring2.p, ring.p + consumed, left * sizeof(float2), hipMemcpyDeviceToDevice));
                ring.swap(ring2);
                ring_len = left;
            }
            if (nf == 0)
                return 0;
            // ---- frame PLL
            if (pll->per_frame() < raw) // with pilots the block leaves the tail of every frame unwritten: defined here
                SD_HIP(hipMemsetAsync(pl.p, 0, (size_t)nf * raw * sizeof(float2), nullptr));
            pll->run(reinterpret_cast<const float *>(fr.p), reinterpret_cast<float *>(pl.p), raw, (int)nf, nullptr);
            st.pll_lanes += pll->stats.lanes;
            st.pll_rerun += pll->stats.rerun;
            st.pll_forced += pll->stats.forced;
            st.pll_serial_frames += pll->stats.serial_frames;
            st.pll_branch_tries += pll->stats.branch_tries;
            // the module's statistics (module_dvbs2_demod.cpp:183-198): the estimate behind every frame, the last one and the peak
            {
                d_m2m4.reserve(2 * (size_t)nf);
                hipLaunchKernelGGL(k_s2_m2m4, dim3((unsigned)nf), dim3(256), 0, nullptr, pl.p, raw, mc.slots * 90, 0.001, d_m2m4.p);
                h_m2m4.resize(2 * (size_t)nf);
                SD_HIP(hipMemcpy(h_m2m4.data(), d_m2m4.p, h_m2m4.size() * sizeof(double), hipMemcpyDeviceToHost));
                for (int64_t f = 0; f < nf; f++)
                {
                    const float y1 = (float)h_m2m4[2 * f], y2 = (float)h_m2m4[2 * f + 1];
                    const float y1_2 = y1 * y1;
                    const float sig = sqrtf(2 * y1_2 - y2), noise = y1 - sqrtf(2 * y1_2 - y2);
                    const float snr = std::max<float>(0, (float)(10.0 * log10(sig / noise)));
                    st.snr = std::isfinite(snr) ? snr : 0.0f;
                    peak_snr = std::max(peak_snr, st.snr);
                }
                st.peak_snr = peak_snr;
            }
            // ---- soft demapper stage, behind the frames that wait for a full decoder group
            const size_t total = pend_frames + (size_t)nf;
            if (total * n_ldpc > soft.cap)
            {
                soft2.reserve(total * n_ldpc);
                if (pend_frames)
                    SD_HIP(hipMemcpy(soft2.p, soft.p, pend_frames * n_ldpc, hipMemcpyDeviceToDevice));
                soft.swap(soft2);
            }
            d_pls.reserve((size_t)nf);
            if (sdhip_s2_bb_to_soft_dev(device, cfg.modcod, cfg.shortframes ? 1 : 0, cfg.pilots ? 1 : 0, reinterpret_cast<const float *>(pl.p), raw, (int)nf, lut_bits.data(),
                                        cfg.lut_resolution, soft.p + pend_frames * n_ldpc, d_pls.p) < 0)
                throw HipError(sdhip_last_error());
            int pls = 0;
            SD_HIP(hipMemcpy(&pls, d_pls.p + (nf - 1), sizeof(int), hipMemcpyDeviceToHost));
            st.detected_modcod = pls >> 2;
            st.detected_shortframes = (pls & 2) ? 1 : 0;
            st.detected_pilots = pls & 1;
            st.plframes += (uint64_t)nf;
            st.pll_freq = pll->state.freq;
            // display_freq = rad_to_hz(current_freq / final_sps, final_samplerate), module_dvbs2_demod.cpp:192
            {
                sdhip_demod_stats ds;
                if (sdhip_demod_get_stats(front, &ds) == 0 && ds.final_sps > 0)
                    st.freq_hz = (float)((current_freq / ds.final_sps) * ds.final_samplerate / (2.0 * M_PI));
            }
            if (cfg.freq_prop_factor != 0.0f && pll->stats.forced * 16 <= pll->stats.lanes)
            { // the hand-over (file header) -- of a loop that HOLDS the stream: while its chain does not certify (acquisition, a stretch of noise: lanes let
              // through as `forced`) its frequency state is not the stream's, and a rotator fed with it runs away and takes the recording with it (measured,
              // visit F of round 5: -45 kHz on a +2.9 kHz offset); the reference's rotator is fed by frames that left S2BBToSoft, i.e. by a locked chain too
                const double g = 1.0 - pow(1.0 - (double)cfg.freq_prop_factor, (double)nf);
                const double d = (double)pll->state.freq * g;
                current_freq -= d;
                pll->add_frequency((float)-d);
                // The symbols still in the ring (what the PL synchroniser has not consumed: up to a frame and more) were turned at the OLD rate and reach the
                // loop in the next call: turned on by exp(-j d i) from the first of them, and the rotator's phase for the next appended symbol moved by the
                // same -d * left, the sum every symbol is turned by is continuous through the hand-over for them too (the loop saw a +d step over
                // the leftover and a -d step behind it)
                if (ring_len)
                {
                    ProfScope _ps("k_s2_rotate", nullptr);
                    hipLaunchKernelGGL(k_s2_rotate, dim3((unsigned)((ring_len + 255) / 256)), dim3(256), 0, nullptr, ring.p, (long long)ring_len, 0.0, -d);
                    rot_phase = fmod(rot_phase - d * (double)ring_len, 6.283185307179586476925);
                }
            }
            // ---- process_s2: whole decoder groups
            const size_t nfull = total / batch * batch;
            pend_frames = total;
            if (nfull == 0)
                return 0;
            d_tr.reserve(nfull / batch);
            if (sdhip_ldpc_decode_dev(ldpc, soft.p, (int)nfull, cfg.ldpc_trials, d_tr.p) < 0)
                throw HipError(sdhip_last_error());
            const int kb = k_ldpc / 8;
            d_pack.reserve(nfull * kb);
            d_corr.reserve(nfull);
            if (sdhip_s2_pack_dev(bch, soft.p, n_ldpc, (int)nfull, d_pack.p, kb) < 0 || sdhip_bch_decode_dev(bch, d_pack.p, (int)nfull, kb, d_corr.p) < 0 ||
                sdhip_bb_descramble_dev(bch, d_pack.p, (int)nfull, kb) < 0)
                throw HipError(sdhip_last_error());
            d_bb.reserve(nfull * (size_t)bbframe_bytes());
            hipLaunchKernelGGL(k_s2_gather_bb, dim3((unsigned)((bbframe_bytes() + 255) / 256), (unsigned)nfull), dim3(256), 0, nullptr, d_pack.p, kb, bbframe_bytes(), (int)nfull, d_bb.p);
            int tr = 0, co = 0;
            SD_HIP(hipMemcpy(&tr, d_tr.p + (nfull / batch - 1), sizeof(int), hipMemcpyDeviceToHost));
            SD_HIP(hipMemcpy(&co, d_corr.p + (nfull - 1), sizeof(int), hipMemcpyDeviceToHost));
            st.ldpc_trials = (float)(tr == -1 ? cfg.ldpc_trials : tr); // module_dvbs2_demod.cpp:254-257
            st.bch_corrections = (float)co;
            st.bbframes += nfull;
            // frames left waiting
            const size_t left = total - nfull;
            if (left)
            {
                soft2.reserve(left * n_ldpc);
                SD_HIP(hipMemcpy(soft2.p, soft.p + nfull * n_ldpc, left * n_ldpc, hipMemcpyDeviceToDevice));
                soft.swap(soft2);
            }
            pend_frames = left;
            return nfull;
        }
        size_t feed_baseband(const void *d_iq, size_t nsamples, int fmt)
        {
            SD_HIP(hipSetDevice(device));
            if (nsamples == 0)
                return 0;
            front_soft.reserve(2 * nsamples + 64);
            symtmp.reserve(nsamples + 64);
            const int64_t ns = sdhip_demod_process_dev(front, d_iq, nsamples, fmt, front_soft.p, 2 * nsamples + 64, reinterpret_cast<float *>(symtmp.p), nsamples + 64, 0);
            if (ns < 0)
                throw HipError(sdhip_last_error());
            st.samples_in += nsamples;
            return feed_symbols(symtmp.p, (size_t)ns / 2);
        }
        DevBuf<float2> symtmp;

        int64_t deliver_dev(size_t nframes, uint8_t *d_out, size_t cap_frames)
        {
            if (nframes > cap_frames)
                throw HipError("dvbs2 demod: BBFRAME output buffer too small");
            if (nframes)
                SD_HIP(hipMemcpy(d_out, d_bb.p, nframes * (size_t)bbframe_bytes(), hipMemcpyDeviceToDevice));
            return (int64_t)nframes;
        }
        void deliver_host(size_t nframes)
        {
            if (!nframes)
                return;
            const size_t nb = nframes * (size_t)bbframe_bytes();
            h_bb.reserve(nb);
            SD_HIP(hipMemcpy(h_bb.p, d_bb.p, nb, hipMemcpyDeviceToHost));
            std::lock_guard<std::mutex> lk(out_mu);
            out_queue.insert(out_queue.end(), h_bb.p, h_bb.p + nb);
        }
        static size_t fmt_bytes(int fmt)
        {
            switch (fmt)
            {
            case SDHIP_FMT_CF32:
            case SDHIP_FMT_CS32:
                return 8;
            case SDHIP_FMT_CS16:
                return 4;
            default:
                return 2;
            }
        }
        std::unique_ptr<HostPipe> pipe;
        std::mutex out_mu;
        // one staged batch: shipped into a device slot by the pipe's first thread, processed by the second
        DevBuf<unsigned char> d_in_slot[2];
        void ship_in(const uint8_t *pinned, size_t bytes, int /*fmt*/, int slot)
        {
            SD_HIP(hipSetDevice(device));
            d_in_slot[slot].reserve(bytes);
            SD_HIP(hipMemcpy(d_in_slot[slot].p, pinned, bytes, hipMemcpyHostToDevice));
        }
        void ship(size_t bytes, int fmt, int slot)
        {
            const size_t ns = bytes / fmt_bytes(fmt);
            if (ns == 0)
                return;
            SD_HIP(hipSetDevice(device));
            deliver_host(feed_baseband(d_in_slot[slot].p, ns, fmt));
        }
        int flush_host()
        {
            if (pipe)
                pipe->flush();
            return 0;
        }
        int push_host(const void *iq, size_t nsamples, int fmt)
        {
            if (!pipe)
                pipe.reset(new HostPipe([this](const uint8_t *p, size_t b, int f, int s) { ship_in(p, b, f, s); }, [this](size_t b, int f, int s) { ship(b, f, s); }));
            pipe->push(iq, nsamples * fmt_bytes(fmt), fmt, HOST_BATCH * fmt_bytes(fmt));
            return 0;
        }
        int64_t pull(uint8_t *out, size_t cap_frames)
        {
            std::lock_guard<std::mutex> lk(out_mu);
            const size_t fb = (size_t)bbframe_bytes();
            const size_t avail = (out_queue.size() - out_read) / fb, take = std::min(avail, cap_frames);
            memcpy(out, out_queue.data() + out_read, take * fb);
            out_read += take * fb;
            if (out_read == out_queue.size())
            {
                out_queue.clear();
                out_read = 0;
            }
            return (int64_t)take;
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
    void sdhip_dvbs2_cfg_default(sdhip_dvbs2_cfg *c)
    {
        memset(c, 0, sizeof(*c));
        sdhip_demod_cfg_default(&c->front);
        // module_dvbs2_demod.h:40-59
        c->front.rrc_taps = 31;
        c->front.clock_gain_omega = (float)(pow(1.7e-3, 2) / 4.0);
        c->front.clock_mu = 0.5f;
        c->front.clock_gain_mu = (float)1.7e-3;
        c->front.clock_omega_relative_limit = 0.005f;
        c->freq_prop_factor = 0.01f;
        c->sof_thresold = 0.6f;
        c->ldpc_trials = 10;
        c->ldpc_batch = 1;
        c->lut_resolution = 256;
    }
    void *sdhip_dvbs2_demod_create(const sdhip_dvbs2_cfg *cfg)
    {
        SD_GUARD_BEGIN
        return new Dvbs2Engine(*cfg);
        SD_GUARD_END(nullptr)
    }
    void sdhip_dvbs2_demod_destroy(void *h) { delete (Dvbs2Engine *)h; }
    int sdhip_dvbs2_demod_bbframe_bytes(void *h) { return ((Dvbs2Engine *)h)->bbframe_bytes(); }
    int sdhip_dvbs2_demod_push(void *h, const void *iq, size_t nsamples, int fmt)
    {
        SD_GUARD_BEGIN
        return ((Dvbs2Engine *)h)->push_host(iq, nsamples, fmt);
        SD_GUARD_END(-1)
    }
    int sdhip_dvbs2_demod_flush(void *h)
    {
        SD_GUARD_BEGIN
        return ((Dvbs2Engine *)h)->flush_host();
        SD_GUARD_END(-1)
    }
    int64_t sdhip_dvbs2_demod_pull(void *h, uint8_t *bbframes, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        return ((Dvbs2Engine *)h)->pull(bbframes, cap_frames);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_dvbs2_demod_process_dev(void *h, const void *d_iq, size_t nsamples, int fmt, uint8_t *d_bbframes, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        Dvbs2Engine *e = (Dvbs2Engine *)h;
        return e->deliver_dev(e->feed_baseband(d_iq, nsamples, fmt), d_bbframes, cap_frames);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_dvbs2_demod_symbols_dev(void *h, const float *d_syms, size_t nsyms, uint8_t *d_bbframes, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        Dvbs2Engine *e = (Dvbs2Engine *)h;
        return e->deliver_dev(e->feed_symbols(reinterpret_cast<const float2 *>(d_syms), nsyms), d_bbframes, cap_frames);
        SD_GUARD_END(-1)
    }
    int sdhip_dvbs2_demod_get_stats(void *h, sdhip_dvbs2_stats *out)
    {
        SD_GUARD_BEGIN
        Dvbs2Engine *e = (Dvbs2Engine *)h;
        std::lock_guard<std::mutex> lk(e->st_mu);
        *out = e->st_busy > 0 ? e->st_pub : e->st;
        return 0;
        SD_GUARD_END(-1)
    }
}
