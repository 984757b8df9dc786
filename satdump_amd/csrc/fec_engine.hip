// fec_engine.hip -- host side of the FEC chain: replays the reference's small sequential state
// machines (Viterbi lock FSM, ASM deframer FSM, MetOp watchdog) around the batch kernels of
// fec_kernels.hip, verifies every speculation the kernels make, and exposes the C ABI of include/sdhip.h.
//
// What runs where:
//   GPU : symbol rotation/conversion, depuncture, ACS, traceback, BER re-encode, lock-search decodes,
//         NRZ-M, exact ASM search over every bit, frame extraction, derandomiser, Reed-Solomon.
//   host: O(#blocks) Viterbi lock FSM (viterbi_1_2.cpp:52-117), O(#frames) deframer FSM
//         (bpsk_ccsds_deframer.cpp:24-107) working on the packed bit stream, rs_usecheck filter.
#include "m2x_deint.h"
#include "fec_kernels.h"
#include "../../include/sdhip.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

namespace sdhip
{
    static thread_local std::string g_last_error;
    void set_error(const std::string &msg) { g_last_error = msg; }

    // ---- block pool behind DevBuf / PinBuf (common.h) ----
    static std::mutex g_pool_mu;
    static bool g_pool_on = false;
    static std::multimap<std::pair<int, size_t>, void *> g_pool_dev; // (device, bytes) -> parked block
    static std::multimap<size_t, void *> g_pool_pin;
    static std::map<void *, int> g_dev_of; // device a live block was allocated on
    void *dev_alloc(size_t bytes)
    {
        int dev = 0;
        SD_HIP(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            auto it = g_pool_dev.find({dev, bytes});
            if (it != g_pool_dev.end())
            {
                void *p = it->second;
                g_pool_dev.erase(it);
                return p;
            }
        }
        void *p = nullptr;
        SD_HIP(hipMalloc(&p, bytes));
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_dev_of[p] = dev;
        return p;
    }
    void dev_free(void *p, size_t bytes)
    {
        if (!p)
            return;
        bool park;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            park = g_pool_on;
        }
        if (park)
        {
            // hipFree waits for the device before it releases a block; a parked block must give the same guarantee, or the next
            // owner (another handle, another stream) could be handed memory that kernels queued by the previous one still read
            // (a DevBuf that grows in the middle of a pass releases its old block while the pass is in flight)
            int cur = 0, own = 0;
            (void)hipGetDevice(&cur);
            {
                std::lock_guard<std::mutex> lk(g_pool_mu);
                own = g_dev_of[p];
            }
            if (own != cur)
                (void)hipSetDevice(own);
            (void)hipDeviceSynchronize();
            if (own != cur)
                (void)hipSetDevice(cur);
            std::lock_guard<std::mutex> lk(g_pool_mu);
            g_pool_dev.insert({{own, bytes}, p});
            return;
        }
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            g_dev_of.erase(p);
        }
        (void)hipFree(p);
    }
    void *pin_alloc(size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            auto it = g_pool_pin.find(bytes);
            if (it != g_pool_pin.end())
            {
                void *p = it->second;
                g_pool_pin.erase(it);
                return p;
            }
        }
        void *p = nullptr;
        SD_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        return p;
    }
    void pin_free(void *p, size_t bytes)
    {
        if (!p)
            return;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            if (g_pool_on)
            {
                g_pool_pin.insert({bytes, p});
                return;
            }
        }
        (void)hipHostFree(p);
    }
    static void pool_trim()
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto &kv : g_pool_dev)
        {
            g_dev_of.erase(kv.second);
            (void)hipFree(kv.second);
        }
        g_pool_dev.clear();
        for (auto &kv : g_pool_pin)
            (void)hipHostFree(kv.second);
        g_pool_pin.clear();
    }

    // ---- per-kernel event timing (process-wide, off by default) ----------------------------------------
    struct ProfRec
    {
        const char *name;
        hipEvent_t a, b;
    };
    static std::mutex g_prof_mu;
    static bool g_prof_on = false;
    static std::vector<ProfRec> g_prof_pending;
    static std::vector<hipEvent_t> g_prof_pool;
    static std::map<std::string, std::pair<double, long long>> g_prof_acc;
    static hipEvent_t prof_event()
    {
        if (!g_prof_pool.empty())
        {
            hipEvent_t e = g_prof_pool.back();
            g_prof_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        SD_HIP(hipEventCreate(&e));
        return e;
    }
    ProfScope::ProfScope(const char *name, hipStream_t stream) : idx(-1), st(stream)
    {
        if (!g_prof_on)
            return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfRec r{name, prof_event(), prof_event()};
        (void)hipEventRecord(r.a, st);
        idx = (int)g_prof_pending.size();
        g_prof_pending.push_back(r);
    }
    ProfScope::~ProfScope()
    {
        if (idx < 0)
            return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        (void)hipEventRecord(g_prof_pending[idx].b, st);
    }
    static void prof_collect()
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for (auto &r : g_prof_pending)
        {
            float ms = 0;
            if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess)
            {
                auto &e = g_prof_acc[r.name];
                e.first += ms;
                e.second += 1;
            }
            g_prof_pool.push_back(r.a);
            g_prof_pool.push_back(r.b);
        }
        g_prof_pending.clear();
    }

    __global__ void k_unpack_bits(const unsigned *vbits, int wpb, int F, int nblk, unsigned char *out)
    {
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= (long long)nblk * F)
            return;
        const int j = (int)(i / F), n = (int)(i % F);
        out[i] = (unsigned char)((vbits[(size_t)j * wpb + (n >> 5)] >> (31 - (n & 31))) & 1u);
    }

    // decoded bits of a block, packed MSB first 32 per word, as the bytes Viterbi27::work's repack loop writes (viterbi27.cpp:43-55)
    __global__ void k_words_to_bytes(const unsigned *vbits, int wpb, int bytes_per_blk, int nblk, unsigned char *out)
    {
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= (long long)nblk * bytes_per_blk)
            return;
        const int j = (int)(i / bytes_per_blk), b = (int)(i % bytes_per_blk);
        out[i] = (unsigned char)((vbits[(size_t)j * wpb + (b >> 2)] >> (24 - 8 * (b & 3))) & 0xFFu);
    }

    // raw (pre NRZ-M) bits [from, from + nbits) of a logical stream -> new carry buffer (word aligned at its start)
    __global__ void k_make_carry(BitStream bs, long long from, int nwords, unsigned *out)
    {
        const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (i >= nwords)
            return;
        // inline copy of stream_raw32 semantics via the public kernel helper is not visible here; re-implemented
        long long g = from + (long long)i * 32;
        unsigned o = 0;
        for (int b = 0; b < 32; b++, g++)
        {
            unsigned bit = 0;
            if (g >= 0 && g < bs.carry_bits)
                bit = (bs.carry[g >> 5] >> (31 - (g & 31))) & 1u;
            else if (g >= bs.carry_bits)
            {
                const long long g2 = g - bs.carry_bits;
                const long long j = g2 / bs.F;
                if (j < bs.nblk)
                {
                    const int n = (int)(g2 - j * bs.F);
                    bit = (bs.vbits[(size_t)j * bs.wpb + (n >> 5)] >> (31 - (n & 31))) & 1u;
                }
            }
            o |= bit << (31 - b);
        }
        out[i] = o;
    }

    struct DeframerState
    {
        int state = 2; // numeric thresholds as in bpsk_ccsds_deframer.h:33-35
        int inv = 0;
        int good = 0, invalid = 0;
        int64_t next_check = 0;     // absolute bit position of the next state-machine evaluation
        int64_t pending_start = -1; // absolute position of the first payload bit of a frame not yet emitted
    };

    struct FecEngine
    {
        sdhip_fec_cfg cfg;
        hipStream_t stream = nullptr;
        int B = 0, F = 0, nber = 0, cadu_bytes = 0, wpb = 0, dstride = 0;
        VitCfg vc{};
        int phases[4] = {0, 0, 0, 0};
        int nphases = 1, n_swap = 1;
        int st_synced = 12, st_syncing = 6;
        double ber_mult = 2.5;
        int max_batch = getenv("SDHIP_FEC_BATCH") ? std::max(1, atoi(getenv("SDHIP_FEC_BATCH"))) : 65536; // blocks per Viterbi launch (decision scratch: 8 B per trellis step, ~2.2 GB at F = 4096)

        // Viterbi FSM (viterbi_1_2.h:25-31)
        int vstate = 0, v_iq_swap = 0, v_phase = 0, v_shift = 0, v_invalid = 0;
        float v_ber = 10;
        float v_bers[2][4][2];
        int dec_first = 1, dec_start = 0; // cc_decoder chaining
        VitSearchState search{};          // cc_decoder_ber / cc_encoder_ber / ber_decoded_buffer
        int metop_nosync_runs = 0;

        // conv_rate != 1/2: viterbi::Viterbi_Depunc (viterbi_punc.{h,cpp}) in front of the same decoder kernels. Host = the reference's
        // state machine call by call (one 8192-symbol block per work()), device = depuncturing, the sliding symbol buffer
        // (ViterbiSlidingBuffer), the decodes and the BER sums. Sequential first cut: coverage of the mode, not speed.
        struct Punc
        {
            int rate = 0;
            PuncPat pat{};
            int is_first = 0, got_extra = 0, changing_shift = 0; // GenericDepunc state (the carried byte lives in d_carry)
            int in_buffer = 0;                                   // ViterbiSlidingBuffer::in_buffer
            int test_bit_len = 0;
            float bers[48];                                      // d_bers[2][12][2], flat (written [s][phase][shift], read [s][o][phase])
            float ber = 10;                                      // d_ber
            int state = 0, iq_swap = 0, phase = 0, shift = 0, invalid = 0;
            int ber_first = 1, ber_start = 0;                    // cc_decoder_ber chaining
            int dec_first = 1, dec_start = 0;                    // cc_decoder chaining
            unsigned enc_state = 0;                              // cc_encoder_ber register
            DevBuf<uint8_t> d_berdep, d_slide, d_tmp, d_carry;
            DevBuf<VitBlockIO> d_io;
            DevBuf<uint64_t> d_dec;
            DevBuf<uint32_t> d_vb_ber;
        } punc;

        // fengyun_ahrpt_decoder (plugins/fengyun3_support/fengyun3/module_fengyun_ahrpt_decoder.cpp:46-126): two Viterbi3_4 (fymode) on the two rails of the
        // QPSK stream, FengyunDiff::work2 on their outputs, then the common deframer / derandomiser / RS tail. Host = the two lock FSMs and the module's
        // counters call by call; device = rail split, lock searches, decodes, BER sums, the differential decoder, everything behind it.
        struct FyRail
        {
            int vstate = 0, v_shift = 0, v_invalid = 0, v_phase = 0; // d_state, d_shift, d_invalid, d_phase (MPT's Viterbi1_2 searches phases 0 / 90)
            float v_ber = 10, bers[4] = {10, 10, 10, 10};            // d_ber, d_bers[0][phase][shift] (fymode never tries phase 1: entries 2, 3 stay 10)
            int dec_first = 1, dec_start = 0;           // cc_decoder chaining
            VitSearchState search{};                    // cc_decoder_ber / cc_encoder_ber
            DevBuf<VitSearchState> d_search;
            DevBuf<VitBlockIO> d_io;
            PinBuf<VitBlockIO> h_io;
            DevBuf<uint32_t> d_vb; // decoded bits, packed
            DevBuf<int8_t> d_rail; // i_soft_buffer / q_soft_buffer of the blocks of one run
            float ber() const { return vstate == 1 ? v_ber : std::min(std::min(10.0f, std::min(bers[0], bers[1])), std::min(bers[2], bers[3])); } // Viterbi3_4::ber(), viterbi_3_4.cpp:173-188 / Viterbi1_2::ber()
        };
        struct Fy
        {
            FyRail rail[2];
            VitCfg rvc{};
            int mpt = 0;                                           // fengyun_mpt_decoder: rate-1/2 rails (Viterbi1_2), see the constructor
            int shift = 0, invert_branches = 0, vit_nosync_run = 0; // the module's `shift`, `invert_branches`, `viterbiNoSyncRun` (`noSyncRuns` = metop_nosync_runs)
            unsigned x_prev = 0, y_prev = 0;                        // FengyunDiff's Xin >> 1, Yin
        } fy;

        // deframer
        DeframerState def;
        int64_t abs_bits = 0; // absolute index of the first bit AFTER everything handed to the deframer so far
        std::vector<uint32_t> carry_host;
        int carry_bits = 0;   // carry holds raw bits [abs_bits - carry_bits, abs_bits)

        // ccsds_simple_psk_decoder runs two deframers on QPSK without NRZ-M (module_ccsds_simple_psk_decoder.cpp:79-80): the
        // members above/below are the ACTIVE deframer context, the other one is parked here (swap_ctx)
        struct DefCtx
        {
            DeframerState def;
            int64_t abs_bits = 0;
            std::vector<uint32_t> carry_host;
            int carry_bits = 0;
            DevBuf<uint32_t> d_carry[2];
            int carry_sel = 0;
            DevBuf<uint32_t> d_vbits;
            DevBuf<uint8_t> d_fbytes;
        } parked;
        void swap_ctx()
        {
            std::swap(def, parked.def);
            std::swap(abs_bits, parked.abs_bits);
            carry_host.swap(parked.carry_host);
            std::swap(carry_bits, parked.carry_bits);
            d_carry[0].swap(parked.d_carry[0]);
            d_carry[1].swap(parked.d_carry[1]);
            std::swap(carry_sel, parked.carry_sel);
            d_vbits.swap(parked.d_vbits);
            d_fbytes.swap(parked.d_fbytes);
        }
        // frames of one deframer over one batch, kept on the device until the caller has merged the two streams' order
        struct FrameBatch
        {
            int nf = 0;
            std::vector<int> keep;         // passes the rs_usecheck filter
            std::vector<int64_t> done_blk; // absolute index of the block in which the reference's deframer->work() returns the frame
        };
        HardCfg hard{};
        long long simple_blocks_done = 0;
        int n_streams = 1;

        // RS bookkeeping
        int last_errors[8] = {0, 0, 0, 0, 0, 0, 0, 0};

        // pending partial block (host copy) and output queue for the host path
        std::vector<int8_t> pending;
        std::vector<uint8_t> out_queue;
        size_t out_queue_read = 0;

        // taps of the last call
        std::vector<float> tap_ber;
        std::vector<int> tap_state;

        sdhip_fec_stats stats{};

        // device buffers
        DevBuf<int8_t> d_stage;
        DevBuf<VitBlockIO> d_io;
        DevBuf<uint64_t> d_dec;
        Vit2Work vit2;
        std::vector<int> redo_list;
        DevBuf<int> d_redo;
        bool use_vit2 = !(getenv("SDHIP_VIT2") && atoi(getenv("SDHIP_VIT2")) == 0);
        DevBuf<uint32_t> d_vbits;
        DevBuf<uint32_t> d_carry[2];
        int carry_sel = 0;
        DevBuf<VitSearchState> d_search;
        DevBuf<uint32_t> d_hits;
        DevBuf<int> d_count;
        DevBuf<uint8_t> d_packed, d_rs_clean;
        DevBuf<uint32_t> d_win;
        PinBuf<uint32_t> h_win;
        unsigned long long stats_full_fetches = 0; // deframer walks that needed the whole packed stream on the host
        DevBuf<FrameDesc> d_frames;
        DevBuf<uint8_t> d_fbytes;
        DevBuf<int> d_ferr;
        DevBuf<int> d_dst;
        DevBuf<uint8_t> d_out_tmp;
        PinBuf<VitBlockIO> h_io;
        PinBuf<uint8_t> h_packed;
        PinBuf<uint32_t> h_hits;
        PinBuf<int> h_ferr, h_dst_pin, h_finfo;
        DevBuf<int> d_finfo, d_fwaves;
        PinBuf<FrameDesc> h_frames_pin;
        std::vector<FrameDesc> h_frames;
        std::vector<int> h_dst;

        explicit FecEngine(const sdhip_fec_cfg &c) : cfg(c)
        {
            SD_HIP(hipSetDevice(cfg.device));
            SD_HIP(hipStreamCreate(&stream));
            if (cfg.cadu_size <= 32 || cfg.cadu_size % 8 != 0)
                throw HipError("cadu_size must be a multiple of 8 bits (padded frames are not supported by the HIP path)");
            cadu_bytes = cfg.cadu_size / 8;
            if (cfg.decoder == SDHIP_DEC_METOP_AHRPT)
            {
                // MetOpAHRPTDecoderModule: BUFFER_SIZE 16384, Viterbi3_4(thr, outsync, 16384), STATE_SYNCED = 18,
                // derand from byte 4, RS223 dual basis I=4 fill 0, every frame written
                // (plugins/noaa_metop_support/metop/module_metop_ahrpt_decoder.cpp:10-28,74-85)
                B = 16384;
                F = 12288;
                nber = 1536;
                ber_mult = 5;
                vc.mode = 1;
                vc.pre_swap = 0;
                nphases = 2;
                phases[0] = 0;
                phases[1] = 1;
                n_swap = 1;
                st_synced = 18;
                cfg.cadu_size = 8192;
                cadu_bytes = 1024;
                cfg.nrzm = 0;
                cfg.derandomize = 1;
                cfg.derand_after_rs = 0;
                cfg.derand_start = 4;
                cfg.rs_i = 4;
                cfg.rs_fill_bytes = 0;
                cfg.rs_dualbasis = 1;
                cfg.rs_type = SDHIP_RS223;
                cfg.rs_usecheck = 0;
                cfg.asm_sync = 0x1ACFFC1D;
            }
            else if (cfg.decoder == SDHIP_DEC_FENGYUN_AHRPT)
            {
                // FengyunAHRPTDecoderModule: BUFFER_SIZE 8192 symbols = 16384 soft bytes per read, Viterbi3_4(thr, outsync, 8192, fymode) per rail,
                // deframer STATE_SYNCING = 8 / STATE_SYNCED = 16, derand from byte 4, RS223 dual basis I=4, every frame written
                // (module_fengyun_ahrpt_decoder.cpp:10-24,46-54,112-121)
                B = 16384;
                F = 12288; // bits per read handed to the deframer: 2 x 6144
                nber = 1536;
                ber_mult = 5;
                vc.mode = 1;
                fy.rvc.mode = 1;
                fy.rvc.fy = 1;
                fy.rvc.B = 8192;
                fy.rvc.F = 6144;
                fy.rvc.nber = 1536;
                for (FyRail &r : fy.rail)
                {
                    memset(&r.search, 0, sizeof(r.search));
                    r.search.ber_first = 1;
                    r.d_search.reserve(1);
                }
                st_syncing = 8;
                st_synced = 16;
                cfg.cadu_size = 8192;
                cadu_bytes = 1024;
                cfg.nrzm = 0;
                cfg.derandomize = 1;
                cfg.derand_after_rs = 0;
                cfg.derand_start = 4;
                cfg.rs_i = 4;
                cfg.rs_fill_bytes = 0;
                cfg.rs_dualbasis = 1;
                cfg.rs_type = SDHIP_RS223;
                cfg.rs_usecheck = 0;
                cfg.asm_sync = 0x1ACFFC1D;
            }
            else if (cfg.decoder == SDHIP_DEC_FENGYUN_MPT)
            {
                // FengyunMPTDecoderModule (module_fengyun_mpt_decoder.cpp:17-31, 43-134): BUFFER_SIZE 8192 symbols = 16384 soft bytes per read, a
                // Viterbi1_2(thr, outsync, 8192, {PHASE_0, PHASE_90}) per rail (4096 bits per rail and read), the deframer as it is constructed (6 / 12),
                // derand from byte 4, RS223 dual basis I=4, every frame written
                B = 16384;
                F = 8192; // bits per read handed to the deframer: 2 x 4096
                nber = 1024;
                ber_mult = 2.5;
                vc.mode = 0;
                fy.mpt = 1;
                fy.rvc.mode = 0;
                fy.rvc.fy = 0;
                fy.rvc.B = 8192;
                fy.rvc.F = 4096;
                fy.rvc.nber = 1024;
                for (FyRail &r : fy.rail)
                {
                    memset(&r.search, 0, sizeof(r.search));
                    r.search.ber_first = 1;
                    r.d_search.reserve(1);
                }
                st_synced = 12;
                cfg.cadu_size = 8192;
                cadu_bytes = 1024;
                cfg.nrzm = 0;
                cfg.derandomize = 1;
                cfg.derand_after_rs = 0;
                cfg.derand_start = 4;
                cfg.rs_i = 4;
                cfg.rs_fill_bytes = 0;
                cfg.rs_dualbasis = 1;
                cfg.rs_type = SDHIP_RS223;
                cfg.rs_usecheck = 0;
                cfg.asm_sync = 0x1ACFFC1D;
            }
            else if (cfg.decoder == SDHIP_DEC_SIMPLE_PSK)
            {
                // CCSDSSimplePSKDecoderModule ctor, module_ccsds_simple_psk_decoder.cpp:19-98: d_buffer_size = d_cadu_size soft bytes
                if (cfg.constellation != SDHIP_BPSK && cfg.constellation != SDHIP_QPSK)
                    throw HipError("CCSDS Simple PSK Decoder : invalid constellation type!");
                if (cfg.rs_i != 0 && cfg.rs_type != SDHIP_RS223 && cfg.rs_type != SDHIP_RS239)
                    throw HipError("CCSDS Simple PSK Decoder : invalid Reed-Solomon type!");
                if (cfg.rs_i < 0 || cfg.rs_i > 8)
                    throw HipError("rs_i out of range");
                B = F = cfg.cadu_size;
                st_synced = 12;
                hard.qpsk = cfg.constellation == SDHIP_QPSK;
                hard.nrzm = cfg.nrzm;
                hard.swap_iq = cfg.qpsk_swap_iq;
                hard.swap_diff = cfg.qpsk_swap_diff;
                hard.oqpsk_delay = cfg.oqpsk_delay;
                hard.method2 = cfg.oqpsk_method2;
                hard.method3 = cfg.oqpsk_method3;
                hard.F = F;
                n_streams = (hard.qpsk && !cfg.nrzm) ? 2 : 1;
                if (hard.qpsk)
                    cfg.nrzm = 0; // QPSK: the differential decoder is part of the bit slicer (QPSKDiff), not NRZ-M on the bit stream
            }
            else
            {
                // CCSDSConvConcatDecoderModule ctor, module_ccsds_conv_concat_decoder.cpp:16-131
                B = std::max(cfg.cadu_size, 8192);
                if (B % 2)
                    throw HipError("odd buffer size");
                F = B / 2;
                nber = 1024;
                ber_mult = 2.5;
                vc.mode = 0;
                if (cfg.conv_rate < 0 || cfg.conv_rate > SDHIP_RATE_7_8)
                    throw HipError("CCSDS Concatenated Decoder : invalid convolutional rate");
                if (cfg.conv_rate != SDHIP_RATE_1_2)
                {
                    punc.rate = cfg.conv_rate;
                    punc.pat = punc_pattern(cfg.conv_rate);
                    for (float &b : punc.bers)
                        b = 10;
                }
                const bool bpsk = cfg.constellation == SDHIP_BPSK || cfg.constellation == SDHIP_BPSK_90;
                const bool bpsk90 = cfg.constellation == SDHIP_BPSK_90;
                if (bpsk && !bpsk90)
                {
                    nphases = 1;
                    phases[0] = 0;
                }
                else if (bpsk90)
                {
                    nphases = 1;
                    phases[0] = 1;
                }
                else if (cfg.constellation == SDHIP_QPSK || cfg.constellation == SDHIP_OQPSK)
                {
                    nphases = 2;
                    phases[0] = 0;
                    phases[1] = 1;
                }
                else
                    throw HipError("CCSDS Concatenated 1/2 Decoder : invalid constellation type!");
                n_swap = cfg.constellation == SDHIP_OQPSK ? 2 : 1;
                vc.pre_swap = (bpsk90 || cfg.iq_invert) ? 1 : 0;
                st_synced = 12;
                if (cfg.rs_i != 0 && cfg.rs_type != SDHIP_RS223 && cfg.rs_type != SDHIP_RS239)
                    throw HipError("CCSDS Concatenated 1/2 Decoder : invalid Reed-Solomon type!");
                if (cfg.rs_i < 0 || cfg.rs_i > 8)
                    throw HipError("rs_i out of range");
                if (cfg.rs_i != 0 && 4 + 255 * cfg.rs_i - std::max(cfg.rs_fill_bytes, 0) * cfg.rs_i > cadu_bytes + 0 && cfg.rs_fill_bytes >= 0 && false)
                    throw HipError("RS block does not fit the CADU");
            }
            vc.B = B;
            vc.F = F;
            vc.nber = nber;
            wpb = cfg.decoder == SDHIP_DEC_SIMPLE_PSK ? (F + 31) / 32 : vit_words_per_block(F);
            dstride = (F + 6 + 63) / 64 * 64;
            for (int s = 0; s < 2; s++)
                for (int p = 0; p < 4; p++)
                    for (int o = 0; o < 2; o++)
                        v_bers[s][p][o] = 10;
            memset(&search, 0, sizeof(search));
            search.ber_first = 1;
            // the deframer's shifter starts at 0: 64 zero bits of history
            carry_bits = 64;
            carry_host.assign(2, 0u);
            def.next_check = 0;
            abs_bits = 0;
            d_search.reserve(1);
            d_count.reserve(1);
            upload_carry();
            if (cfg.m2x_interleaved)
            {
                if (cfg.decoder != SDHIP_DEC_CONV_CONCAT || cfg.constellation != SDHIP_OQPSK || punc.rate != 0 || cfg.cadu_size != 8192)
                    throw HipError("m2x_interleaved: the concatenated decoder's handle with an oqpsk constellation, rate 1/2, 8192-bit CADUs (meteor_lrpt_decoder's m2x_mode)");
                m2x_init();
            }
            if (n_streams == 2)
            { // second deframer: same initial state
                parked.carry_bits = 64;
                parked.carry_host.assign(2, 0u);
                swap_ctx();
                upload_carry();
                swap_ctx();
            }
        }
        ~FecEngine()
        {
            if (stream)
                (void)hipStreamDestroy(stream);
        }

        void upload_carry()
        {
            carry_sel ^= 1;
            DevBuf<uint32_t> &c = d_carry[carry_sel];
            c.reserve(carry_host.size() + 2);
            SD_HIP(hipMemcpyAsync(c.p, carry_host.data(), carry_host.size() * 4, hipMemcpyHostToDevice, stream));
            SD_HIP(hipMemsetAsync(c.p + carry_host.size(), 0, 8, stream));
        }

        // ------------------------------------------------------------------ Viterbi lock search on one block
        void run_search(const int8_t *d_soft, int64_t block)
        {
            SD_HIP(hipMemcpyAsync(d_search.p, &search, sizeof(search), hipMemcpyHostToDevice, stream));
            launch_vit_search(vc, d_soft, block, n_swap, phases, nphases, d_search.p, stream);
            SD_HIP(hipMemcpyAsync(&search, d_search.p, sizeof(search), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            // acceptance rule, viterbi_1_2.cpp:73-86 / viterbi_3_4.cpp:130-143
            v_ber = 10;
            int cand = 0;
            const int nsw = vc.mode == 0 ? n_swap : 1;
            const int nph = vc.mode == 0 ? nphases : 2;
            for (int s = 0; s < nsw; s++)
                for (int pi = 0; pi < nph; pi++)
                    for (int shift = 0; shift < 2; shift++, cand++)
                    {
                        const int phase = vc.mode == 0 ? phases[pi] : pi;
                        const float errors = (float)search.err[cand], total = (float)search.tot[cand];
                        const float ber = (float)((errors / total) * ber_mult);
                        v_bers[s][phase][shift] = ber;
                        if ((v_ber == 10 && ber < cfg.viterbi_ber_thresold) || (v_ber < 10 && ber < v_ber))
                        {
                            v_ber = ber;
                            v_iq_swap = s;
                            vstate = 1;
                            v_phase = phase;
                            v_shift = shift;
                            v_invalid = 0;
                        }
                    }
        }

        float current_ber() const
        { // Viterbi1_2::ber(), viterbi_1_2.cpp:119-133
            if (vstate == 1)
                return v_ber;
            float ber = 10;
            const int nsw = vc.mode == 0 ? n_swap : 1;
            const int nph = vc.mode == 0 ? nphases : 2;
            for (int s = 0; s < nsw; s++)
                for (int pi = 0; pi < nph; pi++)
                    for (int o = 0; o < 2; o++)
                    {
                        const int phase = vc.mode == 0 ? phases[pi] : pi;
                        if (ber > v_bers[s][phase][o])
                            ber = v_bers[s][phase][o];
                    }
            return ber;
        }

        // ------------------------------------------------------------------ deframer FSM on the packed stream
        struct WalkResult
        {
            DeframerState st;
            std::vector<FrameDesc> frames; // pos relative to the call's logical stream
            std::vector<int> state_at;     // deframer state after each block
        };

        static inline uint32_t window_at(const uint8_t *bytes, int64_t p)
        { // 32 bits ending at stream-relative bit p (p >= 31)
            const int64_t s = p - 31;
            const int64_t byte = s >> 3;
            const int sh = (int)(s & 7);
            uint64_t v = 0;
            for (int i = 0; i < 5; i++)
                v = (v << 8) | bytes[byte + i];
            return (uint32_t)((v << (24 + sh)) >> 32);
        }

        // Where the FSM gets its 32-bit windows from. A locked deframer only ever looks at p0 + k * CADU, and a SYNCING one whose
        // check fails looks one and two bits further on before it gives up (three failures -> NOSYNC, which works from the exact-hit
        // list): WIN_OFFS words per frame position were gathered on the device and are all that crossed PCIe (~1 MB instead of the
        // whole packed stream, 67-100 MB per 65536-block batch -- which is what a single bad frame marker used to cost: 7 ms of a
        // MetOp step, whose uncorrectable frames are passed on). Any other position -- a re-lock on another alignment -- makes
        // fetch_full() bring the whole stream over once, and the walk carries on from it.
        struct WindowSource
        {
            const uint32_t *words = nullptr; // gathered windows
            int64_t p0 = 0;                  // stream-relative position of words[0]
            int step = 1, K = 0;
            const uint8_t *bytes = nullptr;  // whole packed stream, once fetched
            std::function<const uint8_t *()> fetch_full;
            int64_t k_last = 0; // the frame position the previous look-up fell on: a locked FSM asks for the next one (no 64-bit divisions on its path: they were
                                // a third of the walk's 4.3 ns per frame, 0.65 ms of host time per MetOp step with the device idle)
            uint32_t at(int64_t p)
            {
                if (!bytes)
                {
                    const int64_t d = p - p0;
                    int64_t off = d - k_last * step;
                    if (off >= step)
                    {
                        k_last++;
                        off -= step;
                    }
                    if (d >= 0 && off >= 0 && off < WIN_OFFS && k_last < K)
                        return words[k_last * WIN_OFFS + off];
                    if (d >= 0 && d % step < WIN_OFFS && d / step < K)
                    {
                        k_last = d / step;
                        return words[(d / step) * WIN_OFFS + d % step];
                    }
                    bytes = fetch_full();
                }
                return window_at(bytes, p);
            }
        };

        // The exact-ASM hit list is only consulted while the FSM is in NOSYNC (a stream start, a loss of lock): it is searched
        // for, copied and sorted on first use -- a locked decoder never pays for it.
        struct HitSource
        {
            std::function<void(std::vector<uint32_t> &)> produce;
            std::vector<uint32_t> hits;
            bool have = false;
            const std::vector<uint32_t> &get()
            {
                if (!have)
                {
                    produce(hits);
                    have = true;
                }
                return hits;
            }
        };

        WalkResult walk(const DeframerState &in, WindowSource &src, int64_t base_abs, int64_t total_rel, HitSource &hs, int nblk) const
        {
            // bytes: NRZ-M decoded logical stream of this call, relative index r <-> absolute base_abs + r
            WalkResult R;
            R.st = in;
            DeframerState &s = R.st;
            R.state_at.assign(nblk, s.state);
            const int CADU = cfg.cadu_size;
            const int64_t avail_end = base_abs + total_rel;
            R.frames.reserve((size_t)(total_rel / CADU + 4));
            const uint32_t ASM = cfg.asm_sync, ASMI = ~cfg.asm_sync;
            int blk_ptr = 0;
            auto blk_end_abs = [&](int j) -> int64_t { return base_abs + carry_bits + (int64_t)(j + 1) * F - 1; };
            auto settle_blocks = [&](int64_t upto_exclusive) {
                while (blk_ptr < nblk && blk_end_abs(blk_ptr) < upto_exclusive)
                    R.state_at[blk_ptr++] = s.state;
            };
            size_t hit_ptr = 0;
            for (;;)
            {
                if (s.pending_start >= 0)
                {
                    const int64_t last = s.pending_start + (CADU - 32) - 1;
                    if (last >= avail_end)
                        break;
                    FrameDesc d;
                    d.pos = s.pending_start - base_abs;
                    d.inv = s.inv;
                    d.pad = 0;
                    R.frames.push_back(d);
                    s.pending_start = -1;
                }
                const int64_t p = s.next_check;
                if (p >= avail_end)
                    break;
                settle_blocks(p);
                if (s.state == 2)
                {
                    // exact ASM / ~ASM, bit by bit (bpsk_ccsds_deframer.cpp:49-67): jump to the next GPU-found hit
                    const int64_t prel = p - base_abs;
                    const std::vector<uint32_t> &hits = hs.get();
                    while (hit_ptr < hits.size() && (int64_t)(hits[hit_ptr] >> 1) < prel)
                        hit_ptr++;
                    if (hit_ptr >= hits.size())
                    {
                        s.next_check = avail_end;
                        break;
                    }
                    const int64_t q = base_abs + (int64_t)(hits[hit_ptr] >> 1);
                    settle_blocks(q);
                    s.inv = (int)(hits[hit_ptr] & 1u);
                    s.state = st_syncing;
                    s.good = s.invalid = 0;
                    s.pending_start = q + 1;
                    s.next_check = q + CADU;
                }
                else
                {
                    const uint32_t w = src.at(p - base_abs);
                    const int dist = __builtin_popcount(w ^ (s.inv ? ASMI : ASM));
                    if (s.state == st_syncing)
                    {
                        if (dist < s.state)
                        {
                            s.pending_start = p + 1;
                            s.next_check = p + CADU;
                            s.invalid = 0;
                            s.good++;
                            if (s.good > 10)
                                s.state = st_synced;
                        }
                        else
                        {
                            s.invalid++;
                            s.good = 0;
                            if (s.invalid > 2)
                                s.state = 2;
                            s.next_check = p + 1;
                        }
                    }
                    else
                    {
                        if (dist < s.state)
                        {
                            s.pending_start = p + 1;
                            s.next_check = p + CADU;
                        }
                        else
                        {
                            s.good = s.invalid = 0;
                            s.state = 2;
                            s.next_check = p + 1;
                        }
                    }
                }
            }
            settle_blocks(avail_end);
            while (blk_ptr < nblk)
                R.state_at[blk_ptr++] = s.state;
            return R;
        }

        // ------------------------------------------------------------------ one run of SYNCED blocks -> frames
        // d_soft: device pointer to block 0 of the contiguous region; [blk0, blk0 + n) were decoded into d_vbits.
        // Returns the number of blocks actually consumed by the deframer (may be < n for the MetOp watchdog).
        // collect != nullptr: do not emit; report the frames (left in d_fbytes) and where the reference would have returned them
        // punctured decoders: the reference call (viterbi.work + ONE deframer.work, module_ccsds_conv_concat_decoder.cpp:93-119) during which each
        // decoded window of the pending batch came out -- one or two windows per call there, so "frames returned by one deframer call" is
        // not "frames ending in one window" 
        std::vector<int64_t> punc_win_call;
        int deframe_and_emit(int n, uint8_t *d_out, size_t out_cap_frames, size_t &out_written, FrameBatch *collect = nullptr)
        {
            const bool tdbg = getenv("SDHIP_DEBUG") != nullptr;
            auto t_prev = std::chrono::steady_clock::now();
            auto tick = [&](const char *what) {
                if (!tdbg)
                    return;
                const auto t = std::chrono::steady_clock::now();
                fprintf(stderr, "[sdhip] fec     . %-12s %7.3f ms (host wall)\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
                t_prev = t;
            };
            int n_eff = n;
            bool watchdog_fired = false;
            for (int attempt = 0; attempt < 3; attempt++)
            {
                BitStream bs;
                bs.carry = d_carry[carry_sel].p;
                bs.carry_bits = carry_bits;
                bs.vbits = d_vbits.p;
                bs.F = F;
                bs.wpb = wpb;
                bs.nblk = n_eff;
                bs.nrzm = cfg.nrzm;
                const int64_t total = carry_bits + (int64_t)n_eff * F;
                const int64_t base_abs = abs_bits - carry_bits;
                // exact hits are only needed from the first position the FSM may evaluate
                const int hits_cap = (int)std::min<int64_t>(total / 64 + 1024, 1 << 26);
                int64_t from = std::max<int64_t>(def.next_check - base_abs, 32);
                const size_t pbytes = (size_t)((total + 31) / 32) * 4 + 8;
                d_packed.reserve(pbytes);
                h_packed.reserve(pbytes);
                launch_pack_stream(bs, d_packed.p, total, stream);
                // the windows a locked FSM will ask for: one per frame from its next check position on
                const int64_t gp0 = def.next_check - base_abs;
                const int gK = (int)std::min<int64_t>(std::max<int64_t>(0, (total - gp0) / cfg.cadu_size + 2), 1 << 24);
                d_win.reserve((size_t)gK * WIN_OFFS + 1);
                h_win.reserve((size_t)gK * WIN_OFFS + 1);
                launch_window_gather(d_packed.p, total, gp0, cfg.cadu_size, gK, d_win.p, stream);
                if (gK > 0)
                    SD_HIP(hipMemcpyAsync(h_win.p, d_win.p, (size_t)gK * WIN_OFFS * 4, hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                HitSource hs;
                hs.produce = [&](std::vector<uint32_t> &hits) {
                    d_hits.reserve(hits_cap);
                    SD_HIP(hipMemsetAsync(d_count.p, 0, sizeof(int), stream));
                    launch_sync_search(bs, from, cfg.asm_sync, d_hits.p, hits_cap, d_count.p, stream);
                    int count = 0;
                    SD_HIP(hipMemcpyAsync(&count, d_count.p, sizeof(int), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    if (count > hits_cap)
                        throw HipError("ASM hit list overflow");
                    hits.resize(count);
                    if (count)
                    {
                        SD_HIP(hipMemcpy(hits.data(), d_hits.p, (size_t)count * 4, hipMemcpyDeviceToHost));
                        for (auto &h : hits)
                            h += (uint32_t)(from << 1);
                        std::sort(hits.begin(), hits.end());
                    }
                };
                tick("search+pack");
                WindowSource src;
                src.words = h_win.p;
                src.p0 = gp0;
                src.step = cfg.cadu_size;
                src.K = gK;
                src.fetch_full = [&]() -> const uint8_t * {
                    SD_HIP(hipMemcpyAsync(h_packed.p, d_packed.p, pbytes - 8, hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    memset(h_packed.p + pbytes - 8, 0, 8);
                    stats_full_fetches++;
                    if (tdbg)
                        fprintf(stderr, "[sdhip] fec     . deframer asked for a window off the gathered grid: whole packed stream fetched (%zu bytes)\n", pbytes);
                    return h_packed.p;
                };
                if (getenv("SDHIP_WINDOW_GATHER") && atoi(getenv("SDHIP_WINDOW_GATHER")) == 0)
                    src.bytes = src.fetch_full(); // A/B switch: the whole stream on the host, as before the gather
                const int st_in = def.state;
                WalkResult W = walk(def, src, base_abs, total, hs, n_eff);
                if (tdbg)
                    fprintf(stderr, "[sdhip] fec     . walk: state %d -> %d, %zu frames, exact-hit list %s (%zu hits), first window at %lld, %d windows x %d\n", st_in, W.st.state,
                            W.frames.size(), hs.have ? "used" : "not needed", hs.hits.size(), (long long)gp0, gK, WIN_OFFS);
                tick("walk");

                if (cfg.decoder == SDHIP_DEC_METOP_AHRPT || cfg.decoder == SDHIP_DEC_FENGYUN_AHRPT || cfg.decoder == SDHIP_DEC_FENGYUN_MPT)
                {
                    // watchdog, module_metop_ahrpt_decoder.cpp:59-72: 10 consecutive calls ending in NOSYNC reset the Viterbi
                    // (module_fengyun_ahrpt_decoder.cpp:97-109: ... exchange the differential decoder's two inputs from the next call on)
                    int runs = metop_nosync_runs, cut = -1;
                    for (int j = 0; j < n_eff; j++)
                    {
                        if (W.state_at[j] == 2)
                        {
                            runs++;
                            if (runs >= 10)
                            {
                                runs = 0;
                                cut = j;
                                break;
                            }
                        }
                        else
                            runs = 0;
                    }
                    if (cut >= 0 && cut + 1 < n_eff)
                    {
                        n_eff = cut + 1; // bits decoded after the reset must not reach the deframer: walk again, shorter
                        continue;
                    }
                    metop_nosync_runs = runs;
                    if (cut >= 0)
                        watchdog_fired = true;
                }

                // ---- frames: extract + derand + RS on the GPU
                const int nf = (int)W.frames.size();
                stats.frames_deframed += nf;
                if (collect)
                {
                    collect->nf = nf;
                    collect->keep.assign(nf, 1);
                    collect->done_blk.resize(nf);
                    for (int f = 0; f < nf; f++)
                        collect->done_blk[f] = (base_abs + W.frames[f].pos + (cfg.cadu_size - 32) - 1) / F;
                }
                if (nf > 1 && cfg.rs_i != 0 && cfg.rs_fill_bytes == -1)
                { // frames the reference's deframer returned from ONE work() call (one Viterbi buffer; one or two on the punctured path): see k_rs_overrun
                    const bool by_call = punc.rate != 0 && (int)punc_win_call.size() >= n_eff;
                    auto call_of = [&](int f) {
                        const int64_t blk = (base_abs + W.frames[f].pos + (cfg.cadu_size - 32) - 1) / F;
                        if (!by_call)
                            return blk;
                        const int64_t w = blk - abs_bits / F; // window of this batch the frame's last bit lies in
                        return (w >= 0 && w < (int64_t)punc_win_call.size()) ? punc_win_call[(size_t)w] : -1 - blk;
                    };
                    for (int f = 0; f + 1 < nf; f++)
                        W.frames[f].pad = call_of(f) == call_of(f + 1) ? 1 : 0;
                }
                if (nf > 0)
                {
                    d_frames.reserve(nf);
                    d_fbytes.reserve((size_t)nf * cadu_bytes + 64);
                    const int I = std::max(cfg.rs_i, 1);
                    d_ferr.reserve((size_t)nf * I);
                    h_ferr.reserve((size_t)nf * I);
                    // (through pinned memory: a std::vector's pages go through the runtime's staging copy, synchronously)
                    h_frames_pin.reserve(nf);
                    memcpy(h_frames_pin.p, W.frames.data(), (size_t)nf * sizeof(FrameDesc));
                    SD_HIP(hipMemcpyAsync(d_frames.p, h_frames_pin.p, (size_t)nf * sizeof(FrameDesc), hipMemcpyHostToDevice, stream));
                    FrameCfg fc;
                    fc.cadu_bits = cfg.cadu_size;
                    fc.cadu_bytes = cadu_bytes;
                    fc.asm_sync = cfg.asm_sync;
                    fc.derand = cfg.derandomize;
                    fc.derand_after_rs = cfg.derand_after_rs;
                    fc.derand_start = cfg.derand_start;
                    fc.rs_i = cfg.rs_i;
                    fc.rs_fill_bytes = cfg.rs_fill_bytes;
                    fc.rs_dualbasis = cfg.rs_dualbasis;
                    fc.rs_nroots = cfg.rs_type == SDHIP_RS239 ? 16 : 32;
                    d_rs_clean.reserve(rs_scratch_bytes((long long)nf * I)); // clean flags, syndromes, dirty list (k_rs_screen -> k_rs)
                    launch_frames(bs, fc, d_frames.p, nf, d_fbytes.p, d_ferr.p, stream, d_rs_clean.p);
                    h_dst.assign(nf, -1);
                    size_t kept = 0;
                    const bool dev_filter = d_out != nullptr && collect == nullptr && !(getenv("SDHIP_RS_FILTER_DEV") && atoi(getenv("SDHIP_RS_FILTER_DEV")) == 0);
                    if (dev_filter)
                    { // device-resident output: the filter and the slots on the device, nine integers back (k_rs_filter)
                        d_dst.reserve(nf);
                        d_finfo.reserve(16);
                        h_finfo.reserve(16);
                        d_fwaves.reserve((size_t)(nf + 255) / 256 * 4 + 4);
                        launch_rs_filter(d_ferr.p, nf, I, cfg.rs_i, cfg.rs_usecheck, (int)out_written, d_dst.p, d_finfo.p, d_fwaves.p, stream);
                        SD_HIP(hipMemcpyAsync(h_finfo.p, d_finfo.p, 9 * sizeof(int), hipMemcpyDeviceToHost, stream));
                        SD_HIP(hipStreamSynchronize(stream));
                        kept = (size_t)h_finfo.p[0];
                        for (int k = 0; k < cfg.rs_i && k < 8; k++)
                            last_errors[k] = h_finfo.p[1 + k];
                        if (kept)
                        {
                            if (out_written + kept > out_cap_frames)
                                throw HipError("CADU output buffer too small");
                            launch_compact(d_fbytes.p, d_dst.p, nf, cadu_bytes, d_out, stream);
                            SD_HIP(hipStreamSynchronize(stream));
                            out_written += kept;
                            stats.frames_out += kept;
                            kept = 0; // (booked)
                        }
                    }
                    else if (cfg.rs_i != 0)
                    {
                        SD_HIP(hipMemcpyAsync(h_ferr.p, d_ferr.p, (size_t)nf * I * sizeof(int), hipMemcpyDeviceToHost, stream));
                        SD_HIP(hipStreamSynchronize(stream));
                    }
                    for (int f = 0; f < nf && !dev_filter; f++)
                    {
                        bool valid = true;
                        if (cfg.rs_i != 0)
                        {
                            for (int k = 0; k < cfg.rs_i; k++)
                            {
                                last_errors[k] = h_ferr.p[(size_t)f * I + k];
                                if (last_errors[k] == -1)
                                    valid = false;
                            }
                        }
                        else
                        { // errors[] keeps its previous content when RS is off (module_ccsds_conv_concat_decoder.cpp:183-186)
                            for (int k = 0; k < 0; k++)
                                (void)k;
                        }
                        if (collect)
                            collect->keep[f] = (!cfg.rs_usecheck || valid) ? 1 : 0;
                        else if (!cfg.rs_usecheck || valid)
                            h_dst[f] = (int)(out_written + kept++);
                    }
                    if (kept)
                    {
                        if (d_out)
                        {
                            if (out_written + kept > out_cap_frames)
                                throw HipError("CADU output buffer too small");
                            d_dst.reserve(nf);
                            h_dst_pin.reserve(nf);
                            memcpy(h_dst_pin.p, h_dst.data(), (size_t)nf * sizeof(int));
                            SD_HIP(hipMemcpyAsync(d_dst.p, h_dst_pin.p, (size_t)nf * sizeof(int), hipMemcpyHostToDevice, stream));
                            launch_compact(d_fbytes.p, d_dst.p, nf, cadu_bytes, d_out, stream);
                            SD_HIP(hipStreamSynchronize(stream));
                        }
                        else
                        {
                            std::vector<uint8_t> tmp((size_t)nf * cadu_bytes);
                            SD_HIP(hipMemcpyAsync(tmp.data(), d_fbytes.p, tmp.size(), hipMemcpyDeviceToHost, stream));
                            SD_HIP(hipStreamSynchronize(stream));
                            for (int f = 0; f < nf; f++)
                                if (h_dst[f] >= 0)
                                    out_queue.insert(out_queue.end(), tmp.begin() + (size_t)f * cadu_bytes, tmp.begin() + (size_t)(f + 1) * cadu_bytes);
                        }
                        out_written += kept;
                        stats.frames_out += kept;
                    }
                }

                tick("frames+rs");
                // ---- commit deframer state and build the next carry (raw bits) ----------------------------
                def = W.st;
                const int64_t avail_end = base_abs + total;
                int64_t keep_from = avail_end - 64;
                if (def.pending_start >= 0)
                    keep_from = std::min(keep_from, def.pending_start - 34);
                keep_from = std::min(keep_from, def.next_check - 34);
                keep_from = std::max(keep_from, base_abs); // cannot keep more than we have
                int new_bits = (int)(avail_end - keep_from);
                const int nwords = (new_bits + 31) / 32;
                const int pad = nwords * 32 - new_bits; // align the END of the carry to a word boundary: pad at the front
                DevBuf<uint32_t> &nc = d_carry[carry_sel ^ 1];
                nc.reserve(nwords + 2);
                hipLaunchKernelGGL(k_make_carry, dim3((nwords + 63) / 64), dim3(64), 0, stream, bs, keep_from - pad - base_abs, nwords, nc.p);
                SD_HIP(hipMemsetAsync(nc.p + nwords, 0, 8, stream));
                SD_HIP(hipStreamSynchronize(stream));
                carry_sel ^= 1;
                carry_bits = nwords * 32;
                abs_bits = avail_end;
                stats.bits_decoded += (uint64_t)n_eff * F;
                if (watchdog_fired)
                    stats.watchdog_events++;
                if (watchdog_fired && (cfg.decoder == SDHIP_DEC_FENGYUN_AHRPT || cfg.decoder == SDHIP_DEC_FENGYUN_MPT))
                    fy.invert_branches ^= 1;
                else if (watchdog_fired)
                    vstate = 0; // viterbi.reset()
                tick("carry");
                return n_eff;
            }
            throw HipError("deframer walk did not converge");
        }

        // ------------------------------------------------------------------ main driver over whole blocks
        // ------------------------------------------------------------------ ccsds_simple_psk_decoder: slicer -> deframer(s) -> derand -> RS
        void emit_frames(const DevBuf<uint8_t> &fbytes, const std::vector<int> &dst, int nf, size_t kept, uint8_t *d_out, size_t out_cap_frames, size_t out_base)
        {
            if (!kept)
                return;
            if (d_out)
            {
                if (out_base + kept > out_cap_frames)
                    throw HipError("CADU output buffer too small");
                d_dst.reserve(nf);
                SD_HIP(hipMemcpyAsync(d_dst.p, dst.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice, stream));
                launch_compact(fbytes.p, d_dst.p, nf, cadu_bytes, d_out, stream);
                SD_HIP(hipStreamSynchronize(stream));
            }
        }
        void process_blocks_simple(const int8_t *d_soft, int64_t nblocks, uint8_t *d_out, size_t out_cap_frames, size_t &out_written)
        {
            int64_t pos = 0;
            while (pos < nblocks)
            {
                const int n = (int)std::min<int64_t>(nblocks - pos, max_batch);
                hard.blocks_done = simple_blocks_done;
                const int8_t *blk = d_soft + pos * (int64_t)B;
                FrameBatch fb[2];
                // stream order of the reference inside one buffer: deframer_qpsk first, then deframer (:185, :264)
                for (int sidx = 0; sidx < n_streams; sidx++)
                {
                    const int which = (n_streams == 2) ? (sidx == 0 ? 1 : 0) : 0;
                    if (sidx == 1)
                        swap_ctx();
                    d_vbits.reserve((size_t)n * wpb + 4);
                    launch_hard_bits(hard, blk, n, which, d_vbits.p, wpb, stream);
                    if (n_streams == 1)
                        deframe_and_emit(n, d_out, out_cap_frames, out_written);
                    else
                        deframe_and_emit(n, d_out, out_cap_frames, out_written, &fb[sidx]);
                    if (sidx == 1)
                        swap_ctx();
                }
                if (n_streams == 2)
                {
                    // merge: per block, deframer_qpsk's frames, then the main deframer's
                    std::vector<int> dst[2] = {std::vector<int>(fb[0].nf, -1), std::vector<int>(fb[1].nf, -1)};
                    size_t i0 = 0, i1 = 0, kept = 0, kept_s[2] = {0, 0};
                    while (i0 < (size_t)fb[0].nf || i1 < (size_t)fb[1].nf)
                    {
                        const bool take0 = i1 >= (size_t)fb[1].nf || (i0 < (size_t)fb[0].nf && fb[0].done_blk[i0] <= fb[1].done_blk[i1]);
                        const int sx = take0 ? 0 : 1;
                        size_t &ix = take0 ? i0 : i1;
                        if (fb[sx].keep[ix])
                        {
                            dst[sx][ix] = (int)(out_written + kept++);
                            kept_s[sx]++;
                        }
                        ix++;
                    }
                    if (d_out)
                    {
                        emit_frames(d_fbytes, dst[0], fb[0].nf, kept_s[0], d_out, out_cap_frames, out_written);
                        emit_frames(parked.d_fbytes, dst[1], fb[1].nf, kept_s[1], d_out, out_cap_frames, out_written);
                    }
                    else if (kept)
                    {
                        std::vector<uint8_t> t0((size_t)fb[0].nf * cadu_bytes), t1((size_t)fb[1].nf * cadu_bytes);
                        if (fb[0].nf)
                            SD_HIP(hipMemcpyAsync(t0.data(), d_fbytes.p, t0.size(), hipMemcpyDeviceToHost, stream));
                        if (fb[1].nf)
                            SD_HIP(hipMemcpyAsync(t1.data(), parked.d_fbytes.p, t1.size(), hipMemcpyDeviceToHost, stream));
                        SD_HIP(hipStreamSynchronize(stream));
                        const size_t base = out_queue.size();
                        out_queue.resize(base + kept * cadu_bytes);
                        for (int sx = 0; sx < 2; sx++)
                            for (int f = 0; f < fb[sx].nf; f++)
                                if (dst[sx][f] >= 0)
                                    memcpy(out_queue.data() + base + (size_t)(dst[sx][f] - (int)out_written) * cadu_bytes,
                                           (sx ? t1 : t0).data() + (size_t)f * cadu_bytes, cadu_bytes);
                    }
                    out_written += kept;
                    stats.frames_out += kept;
                }
                // the two symbols in front of the next call's first soft byte
                if ((size_t)n * B >= 4)
                {
                    int8_t t[4];
                    SD_HIP(hipMemcpyAsync(t, blk + (size_t)n * B - 4, 4, hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    for (int k = 0; k < 4; k++)
                        hard.tail[k] = t[k];
                }
                simple_blocks_done += n;
                pos += n;
                stats.blocks += n;
            }
            stats.deframer_state = std::max(def.state, n_streams == 2 ? parked.def.state : 0);
            for (int k = 0; k < 8; k++)
                stats.rs_errors[k] = last_errors[k];
        }

        // ------------------------------------------------------------------ conv_rate != 1/2 (Viterbi_Depunc)
        float punc_ber() const
        { // Viterbi_Depunc::ber(), viterbi_punc.cpp:151-167 (reads d_bers transposed: [s][o][phase])
            if (punc.state == 1)
                return punc.ber;
            float ber = 10;
            for (int s = 0; s < n_swap; s++)
                for (int pi = 0; pi < nphases; pi++)
                    for (int o = 0; o < 12; o++)
                        if (ber > punc.bers[s * 24 + o * 2 + phases[pi]])
                            ber = punc.bers[s * 24 + o * 2 + phases[pi]];
            return ber;
        }
        // one CCDecoder::work on raw unsigned symbols (cc_decoder.cpp:295-302): frame_bits bits from 2*(frame_bits+6) symbols at d_syms
        VitBlockIO punc_decode(const uint8_t *d_syms, int frame_bits, int &first, int &start, uint32_t *d_vb_out)
        {
            VitCfg v{};
            v.mode = 2;
            v.F = frame_bits;
            v.B = 2 * (frame_bits + 6);
            VitBlockIO one{};
            one.start_in = first ? -2 : start;
            punc.d_io.reserve(1);
            punc.d_dec.reserve((size_t)(frame_bits + 6 + 63) / 64 * 64);
            SD_HIP(hipMemcpyAsync(punc.d_io.p, &one, sizeof(one), hipMemcpyHostToDevice, stream));
            launch_vit_decode(v, (const int8_t *)d_syms, 0, 1, punc.d_io.p, punc.d_dec.p, d_vb_out, stream);
            SD_HIP(hipMemcpyAsync(&one, punc.d_io.p, sizeof(one), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            first = 0;
            start = one.ret_state;
            return one;
        }
        // get_ber over the first nsyms symbols at d_syms against the re-encoded first nsyms/2 bits of d_vb (encoder register enc_in)
        VitBlockIO punc_ber_sums(const uint8_t *d_syms, int frame_bits, int nsyms, const uint32_t *d_vb, unsigned enc_in)
        {
            VitCfg v{};
            v.mode = 2;
            v.F = frame_bits;
            v.B = 2 * (frame_bits + 6);
            v.nber = nsyms / 2;
            VitBlockIO one{};
            launch_vit_ber(v, (const int8_t *)d_syms, 0, 1, d_vb, enc_in, punc.d_io.p, stream);
            SD_HIP(hipMemcpyAsync(&one, punc.d_io.p, sizeof(one), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            return one;
        }
        // ------------------------------------------------------------------ a SYNCED run of punctured blocks in one go
        // The reference feeds a sliding buffer call by call: depuncture the call's symbols to its end, decode a block whenever more
        // than B symbols are in it, drop them, re-encode for the BER, run the lock FSM (viterbi_punc.cpp:103-142,
        // viterbi_buffer.h). The symbol STREAM that passes through the buffer is just the concatenation of what the calls
        // depuncture; the bookkeeping (pattern position, the odd symbol held back so the count stays even, when a block is due, how
        // many symbols lie behind it at that moment) is integer arithmetic on the pattern. So a run of calls is planned on the host
        // (sim), depunctured by ONE launch into a linear buffer, its windows -- B symbols apart, each read 12 symbols into the next
        // -- decoded as one batch by the rate 1/2 engine's own kernels (overlapping blocks: VitCfg::stride) with the start states
        // chained on the device and certified, their BERs computed in one launch, and the lock FSM then walks the calls. What the
        // buffer holds behind its fill level (left-overs the decoder can see when fewer than 12 symbols lie behind a block) is tracked
        // as a short interval list and written back at the end of the run, so the call-by-call path can take over at any call: it
        // does for the lock search, and for the rare call whose block would read into those left-overs.
        struct SlideIv
        {
            int s, e;      // positions [s, e) of the sliding buffer
            int kind;      // 0: what the buffer held at the start of the run (from position off); 1: stream symbols lin[off ..]; 2: 128
            long long off;
        };
        struct PuncWindow
        {
            long long lin_off; // first symbol of the window in the linear stream
            int block;         // call (0-based in the run) during which the reference decodes it
        };
        struct PuncPlan
        {
            int nblk = 0;
            std::vector<PuncDesc> desc;
            std::vector<PuncWindow> win;
            std::vector<SlideIv> map;
            long long lin_len = 0;
            int first_lead = 0;                                        // the first call starts with the carried symbol
            int is_first = 0, got_extra = 0, changing_shift = 0, in_buffer = 0; // depuncturer / buffer state behind the run
        };
        static void iv_set(std::vector<SlideIv> &m, int a, int b, int kind, long long off, int cap)
        {
            b = std::min(b, cap);
            if (a >= b)
                return;
            std::vector<SlideIv> o;
            for (const SlideIv &v : m)
            {
                if (v.e <= a || v.s >= b)
                {
                    o.push_back(v);
                    continue;
                }
                if (v.s < a)
                    o.push_back(SlideIv{v.s, a, v.kind, v.off});
                if (v.e > b)
                    o.push_back(SlideIv{b, v.e, v.kind, v.off + (v.kind == 2 ? 0 : b - v.s)});
            }
            o.push_back(SlideIv{a, b, kind, off});
            std::sort(o.begin(), o.end(), [](const SlideIv &x, const SlideIv &y) { return x.s < y.s; });
            m.swap(o);
        }
        // ViterbiSlidingBuffer::del(n): [n, n + rest) moves to the front, everything from `rest` on stays as it was
        static void iv_del(std::vector<SlideIv> &m, int n, int rest, int cap)
        {
            std::vector<SlideIv> o;
            for (const SlideIv &v : m)
            { // the moved part
                const int a = std::max(v.s, n), b = std::min(v.e, n + rest);
                if (a < b)
                    o.push_back(SlideIv{a - n, b - n, v.kind, v.off + (v.kind == 2 ? 0 : a - v.s)});
            }
            for (const SlideIv &v : m)
            { // what stays
                const int a = std::max(v.s, rest), b = std::min(v.e, cap);
                if (a < b)
                    o.push_back(SlideIv{a, b, v.kind, v.off + (v.kind == 2 ? 0 : a - v.s)});
            }
            std::sort(o.begin(), o.end(), [](const SlideIv &x, const SlideIv &y) { return x.s < y.s; });
            m.swap(o);
        }
        // plan up to nmax calls from the current state; stops in front of a call whose block would be decoded with fewer than 12
        // symbols behind it
        PuncPlan punc_sim(int nmax) const
        {
            const Punc &P = punc;
            const int cap = B * 4 + 64;
            PuncPlan pl;
            pl.map.push_back(SlideIv{0, cap, 0, 0});
            int is_first = P.is_first, got_extra = P.got_extra, cshift = P.changing_shift, in_buffer = P.in_buffer;
            long long consumed = 0, lin_len = in_buffer;
            for (int b = 0; b < nmax; b++)
            {
                // state in front of this call, in case the run has to stop here
                const PuncPlan keep = pl;
                const int k_first = is_first, k_extra = got_extra, k_shift = cshift, k_inb = in_buffer;
                const long long k_cons = consumed, k_len = lin_len;
                const int lead = (is_first || got_extra) ? 1 : 0;
                if (b == 0 && lead)
                {
                    pl.first_lead = 1;
                    lin_len += 1; // the carried symbol is not in the buffer yet: it goes in front of this call's symbols
                }
                is_first = 0;
                got_extra = 0;
                cshift %= P.pat.n;
                const int cnt = punc_count(P.pat, cshift, B);
                pl.desc.push_back(PuncDesc{cshift, lin_len});
                lin_len += cnt;
                cshift += B;
                int sz = lead + cnt;
                if (sz & 1)
                {
                    sz--;
                    got_extra = 1;
                }
                iv_set(pl.map, in_buffer, in_buffer + sz, 1, consumed + in_buffer, cap);
                in_buffer += sz;
                bool rare = false;
                while (in_buffer > B)
                {
                    if (in_buffer - B < 12)
                    {
                        rare = true;
                        break;
                    }
                    pl.win.push_back(PuncWindow{consumed, b});
                    const int rest = in_buffer - B;
                    iv_del(pl.map, B, rest, cap);
                    iv_set(pl.map, rest, rest + 100, 2, 0, cap);
                    consumed += B;
                    in_buffer = rest;
                }
                if (rare)
                { // leave this call to the call-by-call path
                    pl = keep;
                    is_first = k_first;
                    got_extra = k_extra;
                    cshift = k_shift;
                    in_buffer = k_inb;
                    consumed = k_cons;
                    lin_len = k_len;
                    break;
                }
                pl.nblk = b + 1;
            }
            pl.lin_len = lin_len;
            pl.is_first = is_first;
            pl.got_extra = got_extra;
            pl.changing_shift = cshift;
            pl.in_buffer = in_buffer;
            // positions of the map are in buffer coordinates at the END of the run; stream offsets are absolute in lin
            return pl;
        }
        bool punc_batched = true;
        DevBuf<uint8_t> d_punc_lin;
        DevBuf<PuncDesc> d_punc_desc;
        int punc_run(const int8_t *d_soft, int64_t b0, int nmax, VitCfg rot, int &nout)
        {
            const int TEST = 2048;
            Punc &P = punc;
            const int cap = B * 4 + 64;
            PuncPlan pl = punc_sim(nmax);
            if (pl.nblk == 0)
                return 0;
            const int M = (int)pl.win.size();
            // ---- the stream: [what the buffer holds | the carried symbol | the calls' symbols]
            d_punc_lin.reserve((size_t)pl.lin_len + 256);
            if (P.in_buffer > 0)
                SD_HIP(hipMemcpyAsync(d_punc_lin.p, P.d_slide.p, (size_t)P.in_buffer, hipMemcpyDeviceToDevice, stream));
            if (pl.first_lead)
                SD_HIP(hipMemcpyAsync(d_punc_lin.p + P.in_buffer, P.d_carry.p, 1, hipMemcpyDeviceToDevice, stream));
            SD_HIP(hipMemsetAsync(d_punc_lin.p + pl.lin_len, 128, 256, stream));
            d_punc_desc.reserve(pl.desc.size());
            SD_HIP(hipMemcpyAsync(d_punc_desc.p, pl.desc.data(), pl.desc.size() * sizeof(PuncDesc), hipMemcpyHostToDevice, stream));
            rot.iq_swap = P.iq_swap;
            rot.phase = P.phase;
            launch_punc_batch(rot, d_soft, b0, pl.nblk, B, P.pat, d_punc_desc.p, d_punc_lin.p, stream);
            // ---- the windows, as one batch of overlapping blocks
            std::vector<VitBlockIO> io((size_t)std::max(M, 1));
            if (M > 0)
            {
                VitCfg v{};
                v.mode = 2;
                v.F = F;
                v.B = 2 * (F + 6);
                v.stride = B;
                v.nber = P.test_bit_len / 2;
                v.nenc = TEST;
                const int8_t *lin = (const int8_t *)d_punc_lin.p; // window m starts at lin + m * B: block index m of stride B
                d_io.reserve(M);
                h_io.reserve(M);
                for (int j = 0; j < M; j++)
                {
                    h_io.p[j] = VitBlockIO{};
                    h_io.p[j].start_in = -1;
                }
                h_io.p[0].start_in = P.dec_first ? -2 : P.dec_start;
                SD_HIP(hipMemcpyAsync(d_io.p, h_io.p, (size_t)M * sizeof(VitBlockIO), hipMemcpyHostToDevice, stream));
                uint32_t *vb = d_vbits.p + (size_t)nout * wpb;
                const bool v2 = use_vit2 && vit2_supported(v);
                if (!v2)
                    d_dec.reserve((size_t)M * dstride);
                if (v2)
                    launch_vit_decode2(v, lin, 0, M, d_io.p, vb, vit2, stream);
                else
                    launch_vit_decode(v, lin, 0, M, d_io.p, d_dec.p, vb, stream);
                SD_HIP(hipMemcpyAsync(h_io.p, d_io.p, (size_t)M * sizeof(VitBlockIO), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                for (unsigned rounds = 0;; rounds++)
                { // certificates, as in process_blocks: segment certificate of the lane-per-segment kernel, start-state chain
                    redo_list.clear();
                    for (int j = 0; j < M; j++)
                    {
                        if (j > 0 && h_io.p[j].start_used != h_io.p[j - 1].ret_state)
                        {
                            h_io.p[j].start_in = h_io.p[j - 1].ret_state;
                            redo_list.push_back(j);
                            stats.vit_respec++;
                        }
                        else if (h_io.p[j].tb_fallback == 2)
                        {
                            h_io.p[j].start_in = h_io.p[j].start_used;
                            redo_list.push_back(j);
                            stats.tb_respec++;
                        }
                    }
                    if (redo_list.empty())
                        break;
                    if (rounds > (unsigned)M + 2)
                        throw HipError("viterbi start-state chain does not converge");
                    const int nr = (int)redo_list.size();
                    d_redo.reserve(nr);
                    d_dec.reserve((size_t)nr * dstride);
                    SD_HIP(hipMemcpyAsync(d_redo.p, redo_list.data(), (size_t)nr * sizeof(int), hipMemcpyHostToDevice, stream));
                    SD_HIP(hipMemcpyAsync(d_io.p, h_io.p, (size_t)M * sizeof(VitBlockIO), hipMemcpyHostToDevice, stream));
                    launch_vit_decode(v, lin, 0, nr, d_io.p, d_dec.p, vb, stream, d_redo.p);
                    SD_HIP(hipMemcpyAsync(h_io.p, d_io.p, (size_t)M * sizeof(VitBlockIO), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                std::vector<int> rets((size_t)M);
                for (int j = 0; j < M; j++)
                    rets[j] = h_io.p[j].ret_state;
                launch_vit_ber(v, lin, 0, M, vb, P.enc_state, d_io.p, stream);
                SD_HIP(hipMemcpyAsync(h_io.p, d_io.p, (size_t)M * sizeof(VitBlockIO), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                for (int j = 0; j < M; j++)
                {
                    io[j] = h_io.p[j];
                    io[j].ret_state = rets[j];
                }
            }
            // ---- the lock FSM, call by call (viterbi_punc.cpp:126-141)
            int used = pl.nblk, wi = 0, wused = 0;
            for (int b = 0; b < pl.nblk; b++)
            {
                while (wi < M && pl.win[wi].block == b)
                {
                    P.ber = ((float)io[wi].ber_err / (float)io[wi].ber_tot) * 5;
                    wi++;
                }
                wused = wi;
                if (P.ber > cfg.viterbi_ber_thresold)
                {
                    P.invalid++;
                    if ((float)P.invalid > (float)cfg.viterbi_outsync_after)
                        P.state = 0;
                }
                else
                    P.invalid = 0;
                tap_ber.push_back(punc_ber());
                tap_state.push_back(P.state);
                stats.blocks++;
                if (P.state == 0)
                {
                    used = b + 1;
                    break;
                }
            }
            if (used < pl.nblk)
            { // lock lost inside the run: the state behind call `used - 1` is what counts (P's scalars: the FSM ones are already there)
                const float ber = P.ber;
                const int st = P.state, inv = P.invalid;
                // punc_sim reads the depuncturer / buffer scalars only, which this function has not touched yet
                pl = punc_sim(used);
                P.ber = ber;
                P.state = st;
                P.invalid = inv;
            }
            if (wused > 0)
            {
                P.dec_first = 0;
                P.dec_start = io[wused - 1].ret_state;
                P.enc_state = (unsigned)io[wused - 1].pad;
            }
            for (int j = 0; j < wused; j++)
                punc_win_call.push_back(b0 + pl.win[j].block);
            nout += wused;
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] punctured run: %d call(s) planned, %d used, %d block(s) decoded as one batch, lock %s\n", pl.nblk, used, wused,
                        P.state ? "kept" : "lost");
            // ---- hand the buffer back: contents by the interval list, the carried symbol, the scalars
            SD_HIP(hipMemcpyAsync(P.d_tmp.p, P.d_slide.p, (size_t)cap, hipMemcpyDeviceToDevice, stream));
            for (const SlideIv &v : pl.map)
            {
                const size_t n = (size_t)(v.e - v.s);
                if (v.kind == 1)
                    SD_HIP(hipMemcpyAsync(P.d_slide.p + v.s, d_punc_lin.p + v.off, n, hipMemcpyDeviceToDevice, stream));
                else if (v.kind == 2)
                    SD_HIP(hipMemsetAsync(P.d_slide.p + v.s, 128, n, stream));
                else if (v.off != v.s)
                    SD_HIP(hipMemcpyAsync(P.d_slide.p + v.s, P.d_tmp.p + v.off, n, hipMemcpyDeviceToDevice, stream));
            }
            if (pl.got_extra)
                SD_HIP(hipMemcpyAsync(P.d_carry.p, d_punc_lin.p + pl.lin_len - 1, 1, hipMemcpyDeviceToDevice, stream));
            P.is_first = pl.is_first;
            P.got_extra = pl.got_extra;
            P.changing_shift = pl.changing_shift;
            P.in_buffer = pl.in_buffer;
            SD_HIP(hipStreamSynchronize(stream));
            return used;
        }

        void process_blocks_punctured(const int8_t *d_soft, int64_t nblocks, uint8_t *d_out, size_t out_cap_frames, size_t &out_written)
        {
            const int TEST = 2048; // TEST_BITS_LENGTH, viterbi_punc.h:4
            Punc &P = punc;
            if (!P.d_slide.p)
            {
                P.d_berdep.reserve((size_t)TEST * 4 + 64);
                P.d_slide.reserve((size_t)B * 4 + 64);
                P.d_tmp.reserve((size_t)B * 4 + 64);
                P.d_carry.reserve(4);
                P.d_vb_ber.reserve(vit_words_per_block(TEST) + 4);
                // members / heap buffers the reference leaves uninitialised are pinned to zero like in the oracle's wrapper; the
                // depuncturer's carried byte starts as an erasure (depunc.h: buf = 128)
                SD_HIP(hipMemsetAsync(P.d_berdep.p, 0, (size_t)TEST * 4 + 64, stream));
                SD_HIP(hipMemsetAsync(P.d_slide.p, 0, (size_t)B * 4 + 64, stream));
                SD_HIP(hipMemsetAsync(P.d_carry.p, 128, 4, stream));
            }
            tap_ber.reserve(tap_ber.size() + nblocks);
            tap_state.reserve(tap_state.size() + nblocks);
            VitCfg rot = vc; // rotation / swap parameters of SymFetch::u_at
            rot.mode = 0;
            int64_t pos = 0;
            while (pos < nblocks)
            {
                // decoded 4096-bit blocks of this batch go to d_vbits like the rate 1/2 engine's, then to the same deframer
                const int64_t batch = std::min<int64_t>(nblocks - pos, 4096);
                d_vbits.reserve((size_t)(2 * batch + 2) * wpb + 4);
                int nout = 0;
                punc_win_call.clear();
                // one input block the reference's way, call by call: the lock search, and the SYNCED blocks punc_run() leaves alone
                auto seq_block = [&](int64_t b)
                {
                    const int8_t *blk = d_soft + b * (int64_t)B;
                    if (P.state == 0)
                    { // IDLE: viterbi_punc.cpp:55-99
                        P.ber = 10;
                        for (int s = 0; s < n_swap; s++)
                            for (int pi = 0; pi < nphases; pi++)
                            {
                                const int phase = phases[pi];
                                rot.iq_swap = s;
                                rot.phase = phase;
                                for (int shift = 0; shift < P.pat.n * 2; shift++)
                                {
                                    int lenp = (shift > P.pat.n - 1 ? 1 : 0) + punc_count(P.pat, shift % P.pat.n, TEST);
                                    launch_punc_static(rot, blk, P.pat, shift, TEST, P.d_berdep.p, stream);
                                    if (lenp % 2)
                                        lenp--;
                                    punc_decode(P.d_berdep.p, TEST, P.ber_first, P.ber_start, P.d_vb_ber.p);
                                    const VitBlockIO r = punc_ber_sums(P.d_berdep.p, TEST, lenp, P.d_vb_ber.p, P.enc_state);
                                    P.enc_state = (unsigned)r.pad; // the encoder ran lenp/2 bits
                                    P.test_bit_len = lenp;
                                    const float errors = (float)r.ber_err, total = (float)r.ber_tot;
                                    const float ber = (errors / total) * P.pat.berscale;
                                    P.bers[s * 24 + phase * 2 + shift] = ber;
                                    if (ber < cfg.viterbi_ber_thresold && ber < P.ber)
                                    {
                                        P.ber = ber;
                                        P.iq_swap = s;
                                        P.state = 1;
                                        P.phase = phase;
                                        P.shift = shift;
                                        P.invalid = 0;
                                        P.changing_shift = shift; // depunc->set_shift
                                        P.is_first = shift > P.pat.n - 1;
                                    }
                                }
                            }
                    }
                    if (P.state == 1)
                    { // SYNCED: viterbi_punc.cpp:103-142
                        rot.iq_swap = P.iq_swap;
                        rot.phase = P.phase;
                        const int lead = (P.is_first || P.got_extra) ? 1 : 0;
                        P.is_first = 0;
                        P.got_extra = 0;
                        P.changing_shift %= P.pat.n;
                        int sz = lead + punc_count(P.pat, P.changing_shift, B);
                        launch_punc_cont(rot, blk, B, P.pat, P.changing_shift, lead, sz, P.d_carry.p, P.d_slide.p + P.in_buffer, stream);
                        P.changing_shift += B;
                        if (sz % 2)
                        {
                            sz--;
                            P.got_extra = 1;
                        }
                        P.in_buffer += sz; // ViterbiSlidingBuffer::add
                        while (P.in_buffer > B)
                        {
                            uint32_t *vb = d_vbits.p + (size_t)nout * wpb;
                            punc_decode(P.d_slide.p, F, P.dec_first, P.dec_start, vb);
                            // cc_encoder_ber.work(output): TEST bits; get_ber over test_bit_len symbols, scale 5 (viterbi_punc.cpp:126-127)
                            const VitBlockIO r = punc_ber_sums(P.d_slide.p, F, P.test_bit_len, vb, P.enc_state);
                            const float errors = (float)r.ber_err, total = (float)r.ber_tot;
                            P.ber = (errors / total) * 5;
                            uint32_t w[2]; // encoder register after the TEST encoded bits: bits TEST-6 .. TEST-1 of the block
                            SD_HIP(hipMemcpyAsync(w, vb + (TEST - 32) / 32, 4, hipMemcpyDeviceToHost, stream));
                            SD_HIP(hipStreamSynchronize(stream));
                            unsigned e = 0;
                            for (int d = 0; d < 6; d++)
                                e |= ((w[0] >> d) & 1u) << d; // bit TEST-1-d sits at position d of the word that ends at bit TEST-1
                            P.enc_state = e;
                            punc_win_call.push_back(b);
                            nout++;
                            // ViterbiSlidingBuffer::del(B), viterbi_buffer.h:32-37
                            const int rest = P.in_buffer - B;
                            SD_HIP(hipMemcpyAsync(P.d_tmp.p, P.d_slide.p + B, (size_t)rest, hipMemcpyDeviceToDevice, stream));
                            SD_HIP(hipMemcpyAsync(P.d_slide.p, P.d_tmp.p, (size_t)rest, hipMemcpyDeviceToDevice, stream));
                            P.in_buffer = rest;
                            SD_HIP(hipMemsetAsync(P.d_slide.p + P.in_buffer, 128, 100, stream));
                        }
                        if (P.ber > cfg.viterbi_ber_thresold)
                        {
                            P.invalid++;
                            if ((float)P.invalid > (float)cfg.viterbi_outsync_after)
                                P.state = 0;
                        }
                        else
                            P.invalid = 0;
                    }
                    tap_ber.push_back(punc_ber());
                    tap_state.push_back(P.state);
                    stats.blocks++;
                };
                for (int64_t b = pos; b < pos + batch;)
                {
                    if (P.state == 1 && punc_batched)
                    {
                        const int used = punc_run(d_soft, b, (int)(pos + batch - b), rot, nout);
                        if (used > 0)
                        {
                            b += used;
                            continue;
                        }
                    }
                    seq_block(b);
                    b++;
                }
                if (nout > 0)
                    deframe_and_emit(nout, d_out, out_cap_frames, out_written);
                pos += batch;
            }
            stats.viterbi_lock = P.state;
            stats.viterbi_ber = punc_ber();
            stats.deframer_state = def.state;
            for (int k = 0; k < 8; k++)
                stats.rs_errors[k] = last_errors[k];
        }

        // ------------------------------------------------------------------ one run of SYNCED blocks through the decoder
        // Decodes blocks [pos, pos + n) of d_soft under `vc` into d_vbits (packed) and leaves every block's control words -- start / end / chained
        // state, BER sums, encoder register -- in h_io. Speculation (segment warm-ups, start states taken from the previous block's tail) is
        // verified here; what fails is decoded again.
        void vit_run(const VitCfg &vc, const int8_t *d_soft, int64_t pos, int n, int first_start_in, DevBuf<VitBlockIO> &d_io, PinBuf<VitBlockIO> &h_io,
                     DevBuf<uint32_t> &d_vbits, unsigned enc_state_in, const std::function<void(const char *)> &tick)
        {
            const int wpb = vit_words_per_block(vc.F), dstride = (vc.F + 6 + 63) / 64 * 64; // of THIS decoder (the FengYun rails are not the engine's F)
            d_io.reserve(n);
            h_io.reserve(n);
            if (!(use_vit2 && vit2_supported(vc)))
                d_dec.reserve((size_t)n * dstride);
            d_vbits.reserve((size_t)n * wpb + 4);
            for (int j = 0; j < n; j++)
            {
                h_io.p[j] = VitBlockIO{};
                h_io.p[j].start_in = -1;
            }
            h_io.p[0].start_in = first_start_in;
            SD_HIP(hipMemcpyAsync(d_io.p, h_io.p, (size_t)n * sizeof(VitBlockIO), hipMemcpyHostToDevice, stream));
            const bool v2 = use_vit2 && vit2_supported(vc);
            if (v2)
                launch_vit_decode2(vc, d_soft, pos, n, d_io.p, d_vbits.p, vit2, stream);
            else
                launch_vit_decode(vc, d_soft, pos, n, d_io.p, d_dec.p, d_vbits.p, stream);
            // The BER estimate of every block right behind the decode, on the assumption that no block has to be decoded again (the rule by far): ONE copy of the
            // control words and one wait instead of two (3.3 MB and a round trip per 65 536-block batch, with the device idle: 0.28 ms of a MetOp step). A block
            // that fails a certificate below is decoded again and the estimate taken again, as before.
            launch_vit_ber(vc, d_soft, pos, n, d_vbits.p, enc_state_in, d_io.p, stream);
            SD_HIP(hipMemcpyAsync(h_io.p, d_io.p, (size_t)n * sizeof(VitBlockIO), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            bool ber_valid = true;
            // Certificates, in rounds (every failing block of a round is decoded again in ONE launch of the
            // wave-per-block kernel, which is exact within a block given its start state): (1) segment certificate of
            // the lane-per-segment kernel failed (tb_fallback == 2) -> again from the start state it used; (2) the
            // start-state chain (cc_decoder.cpp:295-302): start used != state the previous block's chainback returned.
            unsigned n_cert = 0, n_chain = 0, n_rounds = 0;
            for (;;)
            {
                redo_list.clear();
                for (int j = 0; j < n; j++)
                {
                    if (j > 0 && h_io.p[j].start_used != h_io.p[j - 1].ret_state)
                    {
                        n_chain++;
                        h_io.p[j].start_in = h_io.p[j - 1].ret_state;
                        redo_list.push_back(j);
                    }
                    else if (h_io.p[j].tb_fallback == 2)
                    {
                        n_cert++;
                        h_io.p[j].start_in = h_io.p[j].start_used;
                        redo_list.push_back(j);
                    }
                }
                if (redo_list.empty())
                    break;
                ber_valid = false;
                if (++n_rounds > (unsigned)n + 2)
                    throw HipError("viterbi start-state chain does not converge");
                const int nr = (int)redo_list.size();
                d_redo.reserve(nr);
                d_dec.reserve((size_t)nr * dstride);
                SD_HIP(hipMemcpyAsync(d_redo.p, redo_list.data(), (size_t)nr * sizeof(int), hipMemcpyHostToDevice, stream));
                SD_HIP(hipMemcpyAsync(d_io.p, h_io.p, (size_t)n * sizeof(VitBlockIO), hipMemcpyHostToDevice, stream));
                launch_vit_decode(vc, d_soft, pos, nr, d_io.p, d_dec.p, d_vbits.p, stream, d_redo.p);
                SD_HIP(hipMemcpyAsync(h_io.p, d_io.p, (size_t)n * sizeof(VitBlockIO), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
            }
            stats.vit_respec += n_chain;
            stats.tb_respec += n_cert;
            unsigned n_tbfb = 0;
            for (int j = 0; j < n; j++)
                n_tbfb += h_io.p[j].tb_fallback;
            stats.tb_respec += n_tbfb;
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] viterbi batch %d blocks (%s): segment-certificate re-decodes %u, start-state re-decodes %u in %u round(s), serial tracebacks %u\n", n,
                        v2 ? "lane-per-segment" : "wave-per-block", n_cert, n_chain, n_rounds, n_tbfb);
            tick("viterbi");
            // BER estimate of every block (taken above unless something was decoded again), then the lock FSM (viterbi_1_2.cpp:101-113)
            if (!ber_valid)
            {
                launch_vit_ber(vc, d_soft, pos, n, d_vbits.p, enc_state_in, d_io.p, stream);
                SD_HIP(hipMemcpyAsync(h_io.p, d_io.p, (size_t)n * sizeof(VitBlockIO), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
            }
        }

        // ------------------------------------------------------------------ fengyun_ahrpt_decoder
        // Lock search of one rail's Viterbi3_4 on block 0 of its rail buffer (viterbi_3_4.cpp:110-147, fymode: phase 0, two puncturing shifts)
        void fy_search(FyRail &r)
        {
            SD_HIP(hipMemcpyAsync(r.d_search.p, &r.search, sizeof(r.search), hipMemcpyHostToDevice, stream));
            const int ph[2] = {0, 1};
            const int nph = fy.mpt ? 2 : 1; // MPT: Viterbi1_2 over {PHASE_0, PHASE_90} (viterbi_1_2.cpp:55-90); AHRPT: Viterbi3_4's fymode, phase 0 only
            launch_vit_search(fy.rvc, r.d_rail.p, 0, 1, ph, nph, r.d_search.p, stream);
            SD_HIP(hipMemcpyAsync(&r.search, r.d_search.p, sizeof(r.search), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            r.v_ber = 10;
            int cand = 0;
            for (int pi = 0; pi < nph; pi++)
                for (int shift = 0; shift < 2; shift++, cand++)
                {
                    const float errors = (float)r.search.err[cand], total = (float)r.search.tot[cand];
                    const float ber = (float)((errors / total) * ber_mult);
                    r.bers[2 * pi + shift] = ber;
                    if ((r.v_ber == 10 && ber < cfg.viterbi_ber_thresold) || (r.v_ber < 10 && ber < r.v_ber))
                    {
                        r.v_ber = ber;
                        r.vstate = 1;
                        r.v_phase = ph[pi];
                        r.v_shift = shift;
                        r.v_invalid = 0;
                    }
                }
        }

        // The module's loop (module_fengyun_ahrpt_decoder.cpp:58-126), a run of reads at a time: while either Viterbi is searching, one read per turn
        // (rail split, search, decode of the rail(s) that hold a lock, the counters); once both hold, a batch of reads is decoded for both rails on the
        // assumption that both stay locked, and the two FSMs then walk the batch's BER figures in step -- the read in which either drops out ends the run.
        void process_blocks_fengyun(const int8_t *d_soft, int64_t nblocks, uint8_t *d_out, size_t out_cap_frames, size_t &out_written)
        {
            const int RB = fy.rvc.B, RF = fy.rvc.F, rwpb = vit_words_per_block(RF);
            const auto no_tick = [](const char *) {};
            struct Snap
            {
                int vstate, v_invalid;
                float v_ber;
            };
            std::vector<Snap> snaps[2];
            int64_t pos = 0;
            while (pos < nblocks)
            {
                const bool any_idle = fy.rail[0].vstate == 0 || fy.rail[1].vstate == 0;
                const int n = any_idle ? 1 : (int)std::min<int64_t>(nblocks - pos, max_batch);
                for (FyRail &r : fy.rail)
                    r.d_rail.reserve((size_t)n * RB);
                launch_fy_rails(d_soft, pos, n, fy.shift, cfg.invert_second_viterbi, fy.rail[0].d_rail.p, fy.rail[1].d_rail.p, stream, fy.mpt);
                bool dec[2];
                for (int k = 0; k < 2; k++)
                {
                    FyRail &r = fy.rail[k];
                    if (r.vstate == 0)
                        fy_search(r);
                    dec[k] = r.vstate == 1;
                    if (dec[k])
                    {
                        VitCfg v = fy.rvc;
                        v.shift = r.v_shift;
                        if (fy.mpt)
                            v.phase = r.v_phase;
                        vit_run(v, r.d_rail.p, 0, n, r.dec_first ? -2 : r.dec_start, r.d_io, r.h_io, r.d_vb, r.search.enc_state, no_tick);
                    }
                    snaps[k].assign(n, Snap{r.vstate, r.v_invalid, r.v_ber});
                }
                // the two lock FSMs over the run (viterbi_3_4.cpp:156-168), in step
                int accepted = n;
                for (int j = 0; j < n; j++)
                {
                    for (int k = 0; k < 2; k++)
                    {
                        FyRail &r = fy.rail[k];
                        if (dec[k])
                        {
                            const float errors = (float)r.h_io.p[j].ber_err, total = (float)r.h_io.p[j].ber_tot;
                            r.v_ber = (float)((errors / total) * ber_mult);
                            if (r.v_ber > cfg.viterbi_ber_thresold)
                            {
                                r.v_invalid++;
                                if ((float)r.v_invalid > (float)cfg.viterbi_outsync_after)
                                    r.vstate = 0;
                            }
                            else
                                r.v_invalid = 0;
                        }
                        snaps[k][j] = Snap{r.vstate, r.v_invalid, r.v_ber};
                    }
                    if (fy.rail[0].vstate == 0 || fy.rail[1].vstate == 0)
                    {
                        accepted = j + 1;
                        break;
                    }
                }
                int used = accepted;
                if (dec[0] && dec[1])
                {
                    // v1 > 0 && v2 > 0: differential decoder over the run, then the deframer and what follows it (:93-121)
                    d_vbits.reserve((size_t)accepted * wpb + 4);
                    const FyRail &rx = fy.rail[fy.invert_branches ? 0 : 1], &ry = fy.rail[fy.invert_branches ? 1 : 0];
                    launch_fy_diff(rx.d_vb.p, ry.d_vb.p, accepted, RF, rwpb, fy.x_prev, fy.y_prev, d_vbits.p, wpb, stream);
                    used = deframe_and_emit(accepted, d_out, out_cap_frames, out_written); // < accepted: the branches were exchanged behind read used - 1
                    if (used < accepted)
                        for (int k = 0; k < 2; k++)
                        {
                            fy.rail[k].vstate = snaps[k][used - 1].vstate;
                            fy.rail[k].v_invalid = snaps[k][used - 1].v_invalid;
                            fy.rail[k].v_ber = snaps[k][used - 1].v_ber;
                        }
                    unsigned last[2];
                    SD_HIP(hipMemcpyAsync(&last[0], rx.d_vb.p + (size_t)(used - 1) * rwpb + ((RF - 1) >> 5), 4, hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipMemcpyAsync(&last[1], ry.d_vb.p + (size_t)(used - 1) * rwpb + ((RF - 1) >> 5), 4, hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    fy.x_prev = (last[0] >> (31 - ((RF - 1) & 31))) & 1u;
                    fy.y_prev = (last[1] >> (31 - ((RF - 1) & 31))) & 1u;
                }
                for (int k = 0; k < 2; k++)
                    if (dec[k])
                    {
                        FyRail &r = fy.rail[k];
                        r.dec_first = 0;
                        r.dec_start = r.h_io.p[used - 1].ret_state;
                        r.search.enc_state = (unsigned)r.h_io.p[used - 1].pad;
                    }
                for (int j = 0; j < used; j++)
                    for (int k = 0; k < 2; k++)
                    { // taps: two entries per read, rail 0 then rail 1
                        const Snap &sn = snaps[k][j];
                        tap_ber.push_back(sn.vstate == 1 ? sn.v_ber : std::min(std::min(10.0f, std::min(fy.rail[k].bers[0], fy.rail[k].bers[1])), std::min(fy.rail[k].bers[2], fy.rail[k].bers[3])));
                        tap_state.push_back(sn.vstate);
                    }
                // :80-91, after the two work() calls of a read: only the last read of a run can have left a Viterbi searching (the MPT module tests
                // Viterbi 1's state twice, module_fengyun_mpt_decoder.cpp:78: Viterbi 2 searching alone does not count there)
                if (fy.rail[0].vstate == 0 || (!fy.mpt && fy.rail[1].vstate == 0))
                {
                    fy.vit_nosync_run++;
                    stats.watchdog_events++; // (the counter is cumulative in the module: every counted read is state a cold-started shard does not have)
                    if (fy.vit_nosync_run >= 10)
                        fy.shift ^= 1;
                }
                pos += used;
                stats.blocks += used;
            }
            stats.viterbi_lock = fy.rail[0].vstate;
            stats.viterbi_ber = fy.rail[0].ber();
            stats.viterbi2_lock = fy.rail[1].vstate;
            stats.viterbi2_ber = fy.rail[1].ber();
            stats.deframer_state = def.state;
            for (int k = 0; k < 8; k++)
                stats.rs_errors[k] = last_errors[k];
        }

        void process_blocks(const int8_t *d_soft, int64_t nblocks, uint8_t *d_out, size_t out_cap_frames, size_t &out_written)
        {
            if (cfg.decoder == SDHIP_DEC_FENGYUN_AHRPT || cfg.decoder == SDHIP_DEC_FENGYUN_MPT)
                return process_blocks_fengyun(d_soft, nblocks, d_out, out_cap_frames, out_written);
            if (cfg.decoder == SDHIP_DEC_SIMPLE_PSK)
                return process_blocks_simple(d_soft, nblocks, d_out, out_cap_frames, out_written);
            if (punc.rate != 0)
                return process_blocks_punctured(d_soft, nblocks, d_out, out_cap_frames, out_written);
            int64_t pos = 0;
            tap_ber.reserve(tap_ber.size() + nblocks);
            tap_state.reserve(tap_state.size() + nblocks);
            while (pos < nblocks)
            {
                if (vstate == 0)
                {
                    run_search(d_soft, pos);
                    if (vstate == 0)
                    {
                        tap_ber.push_back(current_ber());
                        tap_state.push_back(0);
                        pos++;
                        stats.blocks++;
                        continue;
                    }
                    stats.viterbi_lock = 1;
                }
                // ---- speculative SYNCED run
                const bool tdbg = getenv("SDHIP_DEBUG") != nullptr;
                auto t_prev = std::chrono::steady_clock::now();
                auto tick = [&](const char *what) {
                    if (!tdbg)
                        return;
                    const auto t = std::chrono::steady_clock::now();
                    fprintf(stderr, "[sdhip] fec   %-10s %7.3f ms (host wall)\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
                    t_prev = t;
                };
                const int n = (int)std::min<int64_t>(nblocks - pos, max_batch);
                vc.iq_swap = v_iq_swap;
                vc.phase = v_phase;
                vc.shift = v_shift;
                vit_run(vc, d_soft, pos, n, dec_first ? -2 : dec_start, d_io, h_io, d_vbits, search.enc_state, tick);
                int accepted = n;
                const size_t tap0 = tap_ber.size();
                for (int j = 0; j < n; j++)
                {
                    const float errors = (float)h_io.p[j].ber_err, total = (float)h_io.p[j].ber_tot;
                    v_ber = (float)((errors / total) * ber_mult);
                    if (v_ber > cfg.viterbi_ber_thresold)
                    {
                        v_invalid++;
                        if ((float)v_invalid > (float)cfg.viterbi_outsync_after)
                            vstate = 0;
                    }
                    else
                        v_invalid = 0;
                    tap_ber.push_back(current_ber());
                    tap_state.push_back(vstate);
                    if (vstate == 0)
                    {
                        accepted = j + 1;
                        break;
                    }
                }
                tick("ber+fsm");
                // hand the accepted blocks to the deframer (the MetOp watchdog may cut the run shorter)
                const int used = deframe_and_emit(accepted, d_out, out_cap_frames, out_written);
                tick("deframe+rs");
                if (used < accepted)
                { // MetOp watchdog reset the Viterbi after block used-1: the rest of the run is decoded again after re-lock
                    accepted = used;
                    tap_ber.resize(tap0 + accepted);
                    tap_state.resize(tap0 + accepted);
                }
                dec_first = 0;
                dec_start = h_io.p[accepted - 1].ret_state;
                search.enc_state = (unsigned)h_io.p[accepted - 1].pad;
                pos += accepted;
                stats.blocks += accepted;
            }
            stats.viterbi_lock = vstate;
            stats.viterbi_ber = current_ber();
            stats.deframer_state = def.state;
            for (int k = 0; k < 8; k++)
                stats.rs_errors[k] = last_errors[k];
        }

        // ------------------------------------------------------------------ meteor_lrpt_decoder, m2x_mode + interleaved (m2x_deint.h has the front)
        // The loop of module_meteor_lrpt_decoder.cpp:130-198 per iteration: deint1 / deint2 ->read_samples (8192 de-interleaved samples each), viterbin / viterbin2
        // ->work on them, the second one's bits if its state is the greater, NRZ-M, deframer, derandomiser, RS. Here: the iterations the data at hand allows, at a time --
        // the two de-interleaved rails of all of them by two gathers (after the autocorrelations have placed the reads), the Viterbis in runs (a locked one decodes its
        // run of reads in one launch; one that is searching searches read by read: its BER decoder is chained from search to search), the selected bits of the run to the
        // deframer in one call.
        struct M2xRail
        {
            int vstate = 0, v_iq_swap = 0, v_phase = 0, v_shift = 0, v_invalid = 0;
            float v_ber = 10, v_bers[2][4][2];
            int dec_first = 1, dec_start = 0;
            VitSearchState search{};
            DevBuf<VitSearchState> d_search;
            DevBuf<VitBlockIO> d_io;
            PinBuf<VitBlockIO> h_io;
            DevBuf<uint32_t> d_vb;
            DevBuf<int8_t> d_rail;
            float ber(int nsw, int nph, const int *ph) const
            { // Viterbi1_2::ber(), viterbi_1_2.cpp:119-133
                if (vstate == 1)
                    return v_ber;
                float b = 10;
                for (int s2 = 0; s2 < nsw; s2++)
                    for (int pi = 0; pi < nph; pi++)
                        for (int o = 0; o < 2; o++)
                            if (b > v_bers[s2][ph[pi]][o])
                                b = v_bers[s2][ph[pi]][o];
                return b;
            }
        };
        struct M2x
        {
            M2xBranch br[2];
            M2xRail rail[2];
            DevBuf<int8_t> d_raw, d_raw2;
            long long raw_base = 0, raw_end = 0; // absolute stream positions of d_raw[0] and of the end of what has been pushed
            bool done = false;                    // the module's loop has ended (should_run() false)
            DevBuf<long long> d_pos;
            DevBuf<int> d_ns, d_res;
            DevBuf<unsigned char> d_hard;
            DevBuf<M2xRead> d_reads;
            std::vector<int> which; // taps: the Viterbi taken per iteration of the last call (1 / 2)
            long long iterations = 0;
        } m2x;

        void m2x_init()
        {
            m2x.br[1].second = 1;
            for (M2xRail &r : m2x.rail)
            {
                memset(&r.search, 0, sizeof(r.search));
                r.search.ber_first = 1;
                r.d_search.reserve(1);
                for (auto &a : r.v_bers)
                    for (auto &b2 : a)
                        for (float &c2 : b2)
                            c2 = 10;
            }
        }
        // append n bytes (device or host memory) to the raw stream kept on the device; bytes no gather can reach any more are dropped first
        void m2x_append(const int8_t *src, size_t n, bool on_device)
        {
            // the oldest call an output of a future call can still come from: 35 x 73 728 data samples = 316 calls back
            long long keep_from = m2x.raw_end;
            for (const M2xBranch &b : m2x.br)
            {
                const long long c_old = std::max<long long>(b.hist_base, b.calls - 320);
                const long long pk = (c_old < b.calls && !b.hist.empty()) ? b.hist[(size_t)(c_old - b.hist_base)].p - 128 : b.p_next - 128;
                keep_from = std::min(keep_from, std::max<long long>(0, pk));
            }
            keep_from = std::max(keep_from, m2x.raw_base);
            const size_t have = (size_t)(m2x.raw_end - m2x.raw_base), live = (size_t)(m2x.raw_end - keep_from);
            if (have + n > m2x.d_raw.cap)
            { // make room: the live part to the front of a buffer that holds it and the new bytes
                m2x.d_raw2.reserve((live + n) * 2 + (1u << 20));
                if (live)
                    SD_HIP(hipMemcpyAsync(m2x.d_raw2.p, m2x.d_raw.p + (keep_from - m2x.raw_base), live, hipMemcpyDeviceToDevice, stream));
                SD_HIP(hipStreamSynchronize(stream));
                m2x.d_raw.swap(m2x.d_raw2);
                m2x.raw_base = keep_from;
            }
            if (n)
                SD_HIP(hipMemcpyAsync(m2x.d_raw.p + (m2x.raw_end - m2x.raw_base), src, n, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
            SD_HIP(hipStreamSynchronize(stream));
            m2x.raw_end += (long long)n;
        }
        // Viterbi1_2's lock search on read `blk` of a rail (viterbi_1_2.cpp:54-92): run_search on the rail's own state
        void m2x_search(M2xRail &r, int64_t blk)
        {
            const bool dbg = getenv("SDHIP_DEBUG") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            SD_HIP(hipMemcpyAsync(r.d_search.p, &r.search, sizeof(r.search), hipMemcpyHostToDevice, stream));
            launch_vit_search(vc, r.d_rail.p, blk, n_swap, phases, nphases, r.d_search.p, stream);
            SD_HIP(hipMemcpyAsync(&r.search, r.d_search.p, sizeof(r.search), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            if (dbg)
                fprintf(stderr, "[sdhip] m2x    search read %lld: %.1f ms\n", (long long)blk, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            r.v_ber = 10;
            int cand = 0;
            for (int s2 = 0; s2 < n_swap; s2++)
                for (int pi = 0; pi < nphases; pi++)
                    for (int shift = 0; shift < 2; shift++, cand++)
                    {
                        const float errors = (float)r.search.err[cand], total = (float)r.search.tot[cand];
                        const float ber = (float)((errors / total) * ber_mult);
                        r.v_bers[s2][phases[pi]][shift] = ber;
                        if ((r.v_ber == 10 && ber < cfg.viterbi_ber_thresold) || (r.v_ber < 10 && ber < r.v_ber))
                        {
                            r.v_ber = ber;
                            r.v_iq_swap = s2;
                            r.vstate = 1;
                            r.v_phase = phases[pi];
                            r.v_shift = shift;
                            r.v_invalid = 0;
                        }
                    }
        }
        void m2x_decode(M2xRail &r, int64_t blk, int n)
        {
            VitCfg v = vc;
            v.iq_swap = r.v_iq_swap;
            v.phase = r.v_phase;
            v.shift = r.v_shift;
            const auto no_tick = [](const char *) {};
            vit_run(v, r.d_rail.p, blk, n, r.dec_first ? -2 : r.dec_start, r.d_io, r.h_io, r.d_vb, r.search.enc_state, no_tick);
        }
        // the BER check behind a decoded read (viterbi_1_2.cpp:101-113)
        void m2x_fsm(M2xRail &r, int j)
        {
            const float errors = (float)r.h_io.p[j].ber_err, total = (float)r.h_io.p[j].ber_tot;
            r.v_ber = (float)((errors / total) * ber_mult);
            if (r.v_ber > cfg.viterbi_ber_thresold)
            {
                r.v_invalid++;
                if ((float)r.v_invalid > (float)cfg.viterbi_outsync_after)
                    r.vstate = 0;
            }
            else
                r.v_invalid = 0;
        }
        void m2x_commit(M2xRail &r, int j)
        { // the decoder objects' chained state behind read j of the rail's current run
            r.dec_first = 0;
            r.dec_start = r.h_io.p[j].ret_state;
            r.search.enc_state = (unsigned)r.h_io.p[j].pad;
        }

        // the Viterbi part of K iterations whose rails are in place; selected bits to the deframer
        void m2x_viterbi(int K, uint8_t *d_out, size_t out_cap_frames, size_t &out_written)
        {
            d_vbits.reserve((size_t)K * wpb + 4);
            int nsel = 0;
            int64_t pos = 0;
            while (pos < K)
            {
                M2xRail *R = m2x.rail;
                bool dec[2];
                for (int k = 0; k < 2; k++)
                {
                    if (R[k].vstate == 0)
                        m2x_search(R[k], pos);
                    dec[k] = R[k].vstate == 1;
                }
                // the run: while a rail is searching it is one read long for that rail's sake unless the other one can carry on (see below)
                const int n = (int)std::min<int64_t>(K - pos, max_batch);
                for (int k = 0; k < 2; k++)
                    if (dec[k])
                        m2x_decode(R[k], pos, n);
                int used = n;
                for (int j = 0; j < n; j++)
                {
                    int out_n[2] = {0, 0};
                    bool ends = false;
                    for (int k = 0; k < 2; k++)
                    {
                        if (dec[k])
                        {
                            out_n[k] = F;
                            m2x_fsm(R[k], j);
                            if (R[k].vstate == 0)
                                ends = true; // it searches again from the next read on
                        }
                        else if (j > 0)
                        { // a searching Viterbi searches every read (read pos was searched above)
                            m2x_search(R[k], pos + j);
                            if (R[k].vstate == 1)
                            { // found: it decodes this very read, and the run ends behind it (from the next read on both decode in step)
                                m2x_decode(R[k], pos + j, 1);
                                out_n[k] = F;
                                m2x_fsm(R[k], 0);
                                m2x_commit(R[k], 0);
                                ends = true;
                            }
                        }
                    }
                    // module_meteor_lrpt_decoder.cpp:145-162
                    const int pick = R[1].vstate > R[0].vstate ? 1 : 0;
                    tap_ber.push_back(R[pick].ber(n_swap, nphases, phases));
                    tap_state.push_back(R[pick].vstate);
                    m2x.which.push_back(pick + 1);
                    if (out_n[pick] > 0)
                    { // (a rail that found its lock at this read decoded it as block 0 of a one-read run of its own)
                        const bool own_run = !dec[pick];
                        SD_HIP(hipMemcpyAsync(d_vbits.p + (size_t)nsel * wpb, R[pick].d_vb.p + (size_t)(own_run ? 0 : j) * wpb, (size_t)wpb * 4, hipMemcpyDeviceToDevice, stream));
                        nsel++;
                    }
                    if (ends || j == n - 1)
                    {
                        used = j + 1;
                        for (int k = 0; k < 2; k++)
                            if (dec[k])
                                m2x_commit(R[k], j);
                        break;
                    }
                }
                pos += used;
                stats.blocks += used;
            }
            if (nsel > 0)
            {
                SD_HIP(hipStreamSynchronize(stream));
                deframe_and_emit(nsel, d_out, out_cap_frames, out_written);
            }
            stats.viterbi_lock = m2x.rail[0].vstate;
            stats.viterbi_ber = m2x.rail[0].ber(n_swap, nphases, phases);
            stats.viterbi2_lock = m2x.rail[1].vstate;
            stats.viterbi2_ber = m2x.rail[1].ber(n_swap, nphases, phases);
            stats.deframer_state = def.state;
            for (int k = 0; k < 8; k++)
                stats.rs_errors[k] = last_errors[k];
        }

        // as many iterations of the module's loop as the stream at hand allows (final: until should_run() turns false)
        void m2x_run(bool final, uint8_t *d_out, size_t out_cap_frames, size_t &out_written)
        {
            const int max_iter = 2048;
            const long long thr = (m2x.raw_end / 8192) * 8192; // final: the fetch that reaches past this is short -> eof (DintSampleReader::read_more, filestream read_data)
            while (!m2x.done)
            {
                // ---- place the next reads of both readers on the assumption that their markers are where they are expected
                int K = max_iter;
                std::vector<long long> pos[2];
                std::vector<int> ns[2];
                for (int b = 0; b < 2; b++)
                {
                    long long p = m2x.br[b].p_next;
                    int k = 0;
                    for (; k < K; k++)
                    {
                        const int nsam = M2xBranch::num_samples(m2x.br[b].calls + k);
                        if (!final && p + nsam + 48 > m2x.raw_end)
                            break;
                        pos[b].push_back(p);
                        ns[b].push_back(nsam);
                        p += nsam;
                    }
                    K = std::min(K, k);
                }
                if (K == 0)
                    break;
                if (getenv("SDHIP_DEBUG"))
                    fprintf(stderr, "[sdhip] m2x    %d reads placed from call %lld (stream %lld .. %lld, final %d)\n", K, m2x.br[0].calls, m2x.raw_base, m2x.raw_end, (int)final);
                // ---- their autocorrelations
                std::vector<int> res[2];
                for (int b = 0; b < 2; b++)
                {
                    m2x.d_pos.reserve(K);
                    m2x.d_ns.reserve(K);
                    m2x.d_res.reserve(2 * (size_t)K);
                    m2x.d_hard.reserve((size_t)K * M2X_HARD_MAX);
                    SD_HIP(hipMemcpyAsync(m2x.d_pos.p, pos[b].data(), (size_t)K * sizeof(long long), hipMemcpyHostToDevice, stream));
                    SD_HIP(hipMemcpyAsync(m2x.d_ns.p, ns[b].data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, stream));
                    {
                        ProfScope _ps("k_m2x_autocorr", stream);
                        hipLaunchKernelGGL(k_m2x_autocorr, dim3((K + 63) / 64), dim3(64), 0, stream, (const signed char *)m2x.d_raw.p, m2x.raw_base, m2x.raw_end, b, m2x.d_pos.p, m2x.d_ns.p, K,
                                           m2x.d_hard.p, m2x.d_res.p);
                    }
                    res[b].resize(2 * (size_t)K);
                    SD_HIP(hipMemcpyAsync(res[b].data(), m2x.d_res.p, res[b].size() * sizeof(int), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                // ---- the calls that stand: up to and including the first one of either reader whose marker was NOT where it was expected (what follows it moves)
                int Kc = K;
                for (int b = 0; b < 2; b++)
                    for (int k = 0; k < Kc; k++)
                        if (M2xBranch::offset_of(m2x.br[b].calls + k, res[b][2 * (size_t)k]) != 0)
                        {
                            Kc = k + 1;
                            break;
                        }
                if (final)
                    for (int k = 0; k < Kc; k++)
                    { // the iteration in which a fetch comes up short is the last one
                        long long demand = 0;
                        for (int b = 0; b < 2; b++)
                            demand = std::max(demand, pos[b][k] + ns[b][k] + std::max(0, M2xBranch::offset_of(m2x.br[b].calls + k, res[b][2 * (size_t)k])));
                        if (demand > thr)
                        {
                            Kc = k + 1;
                            m2x.done = true;
                            break;
                        }
                    }
                // ---- descriptors, rails
                const long long c0 = m2x.br[0].calls; // (both readers have made the same number of calls)
                for (int b = 0; b < 2; b++)
                {
                    M2xBranch &br = m2x.br[b];
                    for (int k = 0; k < Kc; k++)
                    {
                        const int off = M2xBranch::offset_of(br.calls + k, res[b][2 * (size_t)k]);
                        br.hist.push_back(M2xRead{pos[b][k], off, res[b][2 * (size_t)k + 1], ns[b][k], 0});
                        br.rotation = res[b][2 * (size_t)k + 1];
                        br.p_next = pos[b][k] + ns[b][k] + off;
                    }
                    br.calls += Kc;
                    if (br.hist.size() > 4096)
                    { // calls no output can come from any more
                        const size_t drop = br.hist.size() - 1024;
                        br.hist.erase(br.hist.begin(), br.hist.begin() + drop);
                        br.hist_base += (long long)drop;
                    }
                    const long long rb = std::max<long long>(br.hist_base, c0 - 320);
                    const int nr = (int)(br.calls - rb);
                    m2x.d_reads.reserve(nr);
                    SD_HIP(hipMemcpyAsync(m2x.d_reads.p, br.hist.data() + (rb - br.hist_base), (size_t)nr * sizeof(M2xRead), hipMemcpyHostToDevice, stream));
                    m2x.rail[b].d_rail.reserve((size_t)Kc * M2X_LEN + 64);
                    {
                        ProfScope _ps("k_m2x_gather", stream);
                        hipLaunchKernelGGL(k_m2x_gather, dim3((unsigned)(((size_t)Kc * M2X_LEN + 255) / 256)), dim3(256), 0, stream, (const signed char *)m2x.d_raw.p, m2x.raw_base,
                                           m2x.raw_end, b, m2x.d_reads.p, rb, nr, c0, Kc, (signed char *)m2x.rail[b].d_rail.p);
                    }
                    SD_HIP(hipStreamSynchronize(stream)); // (d_reads is reused by the other reader)
                }
                m2x.iterations += Kc;
                if (getenv("SDHIP_DEBUG"))
                    fprintf(stderr, "[sdhip] m2x    %d reads stand (offsets %d / %d, rotations %d / %d); next windows at %lld / %lld\n", Kc, m2x.br[0].hist.back().off, m2x.br[1].hist.back().off,
                            m2x.br[0].rotation, m2x.br[1].rotation, m2x.br[0].p_next, m2x.br[1].p_next);
                m2x_viterbi(Kc, d_out, out_cap_frames, out_written);
                if (getenv("SDHIP_DEBUG"))
                    fprintf(stderr, "[sdhip] m2x    viterbi states %d / %d, deframer %d\n", m2x.rail[0].vstate, m2x.rail[1].vstate, def.state);
            }
        }

        // ------------------------------------------------------------------ entry points
        int64_t process_dev(const int8_t *d_soft, size_t n, uint8_t *d_out, size_t out_cap_frames)
        {
            SD_HIP(hipSetDevice(cfg.device));
            tap_ber.clear();
            tap_state.clear();
            size_t out_written = 0;
            size_t off = 0;
            stats.soft_in += n;
            if (cfg.m2x_interleaved)
            {
                m2x.which.clear();
                m2x_append(d_soft, n, true);
                m2x_run(false, d_out, out_cap_frames, out_written);
                return (int64_t)out_written;
            }
            if (!pending.empty())
            {
                const size_t need = (size_t)B - pending.size();
                const size_t take = std::min(need, n);
                const size_t old = pending.size();
                pending.resize(old + take);
                SD_HIP(hipMemcpy(pending.data() + old, d_soft, take, hipMemcpyDeviceToHost));
                off = take;
                if (pending.size() == (size_t)B)
                {
                    d_stage.reserve(B);
                    SD_HIP(hipMemcpy(d_stage.p, pending.data(), B, hipMemcpyHostToDevice));
                    pending.clear();
                    process_blocks(d_stage.p, 1, d_out, out_cap_frames, out_written);
                }
            }
            const size_t nb = (n - off) / B;
            if (nb)
                process_blocks(d_soft + off, (int64_t)nb, d_out, out_cap_frames, out_written);
            off += nb * (size_t)B;
            if (off < n)
            {
                const size_t old = pending.size();
                pending.resize(old + (n - off));
                SD_HIP(hipMemcpy(pending.data() + old, d_soft + off, n - off, hipMemcpyDeviceToHost));
            }
            return (int64_t)out_written;
        }

        DevBuf<int8_t> d_push;
        int64_t flush(uint8_t *d_out, size_t out_cap_frames)
        {
            SD_HIP(hipSetDevice(cfg.device));
            size_t out_written = 0;
            if (cfg.m2x_interleaved && !m2x.done)
            {
                tap_ber.clear();
                tap_state.clear();
                m2x.which.clear();
                m2x_run(true, d_out, out_cap_frames, out_written);
                m2x.done = true;
            }
            return (int64_t)out_written;
        }
        int push_host(const int8_t *soft, size_t n)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (cfg.m2x_interleaved)
            {
                stats.soft_in += n;
                size_t out_written = 0;
                tap_ber.clear();
                tap_state.clear();
                m2x.which.clear();
                m2x_append(soft, n, false);
                m2x_run(false, nullptr, 0, out_written);
                return 0;
            }
            // assemble pending + new data on the host, ship whole blocks
            const size_t old = pending.size();
            pending.insert(pending.end(), soft, soft + n);
            stats.soft_in += n;
            const size_t nb = pending.size() / B;
            (void)old;
            if (nb)
            {
                d_push.reserve(nb * (size_t)B);
                SD_HIP(hipMemcpy(d_push.p, pending.data(), nb * (size_t)B, hipMemcpyHostToDevice));
                size_t out_written = 0;
                tap_ber.clear();
                tap_state.clear();
                process_blocks(d_push.p, (int64_t)nb, nullptr, 0, out_written);
                pending.erase(pending.begin(), pending.begin() + nb * (size_t)B);
            }
            return 0;
        }
        int64_t pull(uint8_t *cadu, size_t cap_frames)
        {
            const size_t avail = (out_queue.size() - out_queue_read) / cadu_bytes;
            const size_t take = std::min(avail, cap_frames);
            memcpy(cadu, out_queue.data() + out_queue_read, take * cadu_bytes);
            out_queue_read += take * cadu_bytes;
            if (out_queue_read == out_queue.size())
            {
                out_queue.clear();
                out_queue_read = 0;
            }
            return (int64_t)take;
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

namespace sdhip
{
    // viterbi::Viterbi27::work over nframes consecutive calls of one decoder (viterbi27.cpp:31-66), everything on the device: frame j reads
    // d_soft + j * 2 * frame_bits signed soft symbols, d_out gets frame_bits / 8 bytes per frame. start_in0 = -2: the decoder's first call ever
    // (unbiased metrics), else the chained start state the previous call of this decoder returned (*ret_state). ber_err[j] = the numerator of
    // Viterbi27::ber() after frame j (x 4 / ber_test_size). Current device, default stream, synchronous.
    void viterbi27_frames(int frame_bits, int ber_test_size, const int8_t *d_soft, int nframes, int start_in0, uint8_t *d_out, std::vector<int> *ber_err, int *ret_state,
                          unsigned *enc_state)
    {
        if (frame_bits < 64 || frame_bits % 32 || ber_test_size < 2 || ber_test_size % 2 || ber_test_size > 2 * frame_bits)
            throw HipError("viterbi27: frame_bits must be a multiple of 32 and ber_test_size even, <= 2 * frame_bits");
        if (nframes <= 0)
            return;
        VitCfg vc{};
        vc.mode = 0; // signed soft symbols, no rotation: utils.cpp:3-12
        vc.F = frame_bits;
        vc.B = 2 * frame_bits;
        vc.nber = ber_test_size / 2;
        vc.nenc = ber_test_size / 2;
        const int wpb = vit_words_per_block(frame_bits);
        const int dstride = (frame_bits + 6 + 63) / 64 * 64;
        DevBuf<VitBlockIO> d_io;
        DevBuf<uint64_t> d_dec;
        DevBuf<uint32_t> d_vb;
        d_io.reserve(nframes);
        d_vb.reserve((size_t)nframes * wpb + 4);
        std::vector<VitBlockIO> io(nframes);
        for (int j = 0; j < nframes; j++)
            io[j].start_in = -1;
        io[0].start_in = start_in0;
        SD_HIP(hipMemcpy(d_io.p, io.data(), io.size() * sizeof(VitBlockIO), hipMemcpyHostToDevice));
        Vit2Work vit2;
        const bool v2 = vit2_supported(vc) && !(getenv("SDHIP_VIT2") && atoi(getenv("SDHIP_VIT2")) == 0);
        if (v2)
            launch_vit_decode2(vc, d_soft, 0, nframes, d_io.p, d_vb.p, vit2, nullptr);
        else
        {
            d_dec.reserve((size_t)nframes * dstride);
            launch_vit_decode(vc, d_soft, 0, nframes, d_io.p, d_dec.p, d_vb.p, nullptr);
        }
        SD_HIP(hipMemcpy(io.data(), d_io.p, io.size() * sizeof(VitBlockIO), hipMemcpyDeviceToHost));
        d_dec.reserve((size_t)dstride);
        auto redo = [&](int j, int start) {
            VitBlockIO one{};
            one.start_in = start;
            SD_HIP(hipMemcpy(d_io.p + j, &one, sizeof(one), hipMemcpyHostToDevice));
            launch_vit_decode(vc, d_soft, j, 1, d_io.p + j, d_dec.p, d_vb.p + (size_t)j * wpb, nullptr);
            SD_HIP(hipMemcpy(&io[j], d_io.p + j, sizeof(one), hipMemcpyDeviceToHost));
        };
        for (int j = 0; j < nframes; j++)
        {
            if (io[j].tb_fallback == 2)
                redo(j, io[j].start_used);
            if (j > 0 && io[j].start_used != io[j - 1].ret_state)
                redo(j, io[j - 1].ret_state);
        }
        // (the BER re-encoder's register carries over from call to call like CCEncoder::work's d_start_state does across Viterbi27::work calls: ADVICE r4)
        launch_vit_ber(vc, d_soft, 0, nframes, d_vb.p, enc_state ? *enc_state : 0u, d_io.p, nullptr);
        const long long nbytes = (long long)nframes * (frame_bits / 8);
        hipLaunchKernelGGL(k_words_to_bytes, dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, nullptr, d_vb.p, wpb, frame_bits / 8, nframes, d_out);
        SD_HIP(hipMemcpy(io.data(), d_io.p, io.size() * sizeof(VitBlockIO), hipMemcpyDeviceToHost));
        if (ber_err)
        {
            ber_err->resize(nframes);
            for (int j = 0; j < nframes; j++)
                (*ber_err)[j] = io[j].ber_err;
        }
        if (ret_state)
            *ret_state = io[nframes - 1].ret_state;
        if (enc_state)
            *enc_state = (unsigned)io[nframes - 1].pad;
        SD_HIP(hipDeviceSynchronize());
    }
} // namespace sdhip

extern "C"
{
    const char *sdhip_last_error(void) { return g_last_error.c_str(); }
    const char *sdhip_version(void) { return "sdhip 0.1 (gfx950)"; }
    int sdhip_device_count(void)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess)
            return 0;
        return n;
    }

    void sdhip_pool_enable(int on)
    {
        {
            std::lock_guard<std::mutex> lk(sdhip::g_pool_mu);
            sdhip::g_pool_on = on != 0;
        }
        if (!on)
            sdhip::pool_trim();
    }
    void sdhip_pool_trim(void) { sdhip::pool_trim(); }

    void sdhip_prof_enable(int on)
    {
        prof_collect();
        g_prof_on = on != 0;
    }
    void sdhip_prof_reset(void)
    {
        prof_collect();
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_acc.clear();
    }
    int sdhip_prof_get(int idx, char *name, size_t name_cap, double *total_ms, long long *launches)
    {
        prof_collect();
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (idx >= 0 && idx < (int)g_prof_acc.size() && name && name_cap)
        {
            auto it = g_prof_acc.begin();
            std::advance(it, idx);
            snprintf(name, name_cap, "%s", it->first.c_str());
            if (total_ms)
                *total_ms = it->second.first;
            if (launches)
                *launches = it->second.second;
        }
        return (int)g_prof_acc.size();
    }

    void sdhip_fec_cfg_default(sdhip_fec_cfg *c)
    {
        memset(c, 0, sizeof(*c));
        c->decoder = SDHIP_DEC_CONV_CONCAT;
        c->constellation = SDHIP_BPSK;
        c->cadu_size = 8192;
        c->viterbi_outsync_after = 20;
        c->viterbi_ber_thresold = 0.3f;
        c->derandomize = 1;
        c->derand_start = 4;
        c->rs_i = 0;
        c->rs_fill_bytes = -1;
        c->rs_dualbasis = 1;
        c->rs_type = SDHIP_RS_NONE;
        c->asm_sync = 0x1ACFFC1Du;
        c->qpsk_swap_diff = 1;
    }

    void *sdhip_fec_create(const sdhip_fec_cfg *cfg)
    {
        SD_GUARD_BEGIN
        return new FecEngine(*cfg);
        SD_GUARD_END(nullptr)
    }
    void sdhip_fec_destroy(void *h) { delete (FecEngine *)h; }
    int sdhip_fec_push(void *h, const int8_t *soft, size_t n)
    {
        SD_GUARD_BEGIN
        return ((FecEngine *)h)->push_host(soft, n);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_fec_pull(void *h, uint8_t *cadu, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        return ((FecEngine *)h)->pull(cadu, cap_frames);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_fec_flush(void *h, uint8_t *d_cadu, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        return ((FecEngine *)h)->flush(d_cadu, cap_frames);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_fec_process_dev(void *h, const int8_t *d_soft, size_t n, uint8_t *d_cadu, size_t cap_frames)
    {
        SD_GUARD_BEGIN
        return ((FecEngine *)h)->process_dev(d_soft, n, d_cadu, cap_frames);
        SD_GUARD_END(-1)
    }
    int sdhip_fec_get_stats(void *h, sdhip_fec_stats *st)
    {
        *st = ((FecEngine *)h)->stats;
        return 0;
    }
    int64_t sdhip_fec_get_block_taps(void *h, float *blk_ber, int *blk_state, size_t cap)
    {
        FecEngine *e = (FecEngine *)h;
        const size_t n = std::min(cap, e->tap_ber.size());
        if (blk_ber)
            memcpy(blk_ber, e->tap_ber.data(), n * sizeof(float));
        if (blk_state)
            memcpy(blk_state, e->tap_state.data(), n * sizeof(int));
        return (int64_t)e->tap_ber.size();
    }

    // viterbi::Viterbi27::work over nframes consecutive calls (viterbi27.cpp:31-66: signed_soft_to_unsigned, CCDecoder::work with the 12
    // erasure symbols hard_buffer keeps behind the frame, MSB-first repack, re-encode BER x 4 over ber_test_size symbols), CCSDS polys
    int sdhip_op_viterbi27(int device, int frame_bits, int ber_test_size, const int8_t *d_soft, int nframes, uint8_t *d_out, float *ber_out)
    {
        SD_GUARD_BEGIN
        SD_HIP(hipSetDevice(device));
        if (nframes <= 0)
            return 0;
        std::vector<int> err;
        int ret = -2;
        sdhip::viterbi27_frames(frame_bits, ber_test_size, d_soft, nframes, -2, d_out, &err, &ret, nullptr);
        if (ber_out)
            for (int j = 0; j < nframes; j++)
                ber_out[j] = ((float)err[j] / (float)ber_test_size) * 4.0f;
        return 0;
        SD_GUARD_END(-1)
    }

    int sdhip_op_ccdecoder(int device, int frame_bits, const uint8_t *d_syms, int nblocks, uint8_t *d_out)
    {
        SD_GUARD_BEGIN
        SD_HIP(hipSetDevice(device));
        VitCfg vc{};
        vc.mode = 2; // raw unsigned symbols, 2*(F+6) per block, tail included
        vc.F = frame_bits;
        vc.B = 2 * (frame_bits + 6);
        vc.nber = 0;
        const int wpb = vit_words_per_block(frame_bits);
        const int dstride = (frame_bits + 6 + 63) / 64 * 64;
        DevBuf<VitBlockIO> d_io;
        DevBuf<uint64_t> d_dec;
        DevBuf<uint32_t> d_vb;
        d_io.reserve(nblocks);
        d_dec.reserve((size_t)nblocks * dstride);
        d_vb.reserve((size_t)nblocks * wpb + 4);
        std::vector<VitBlockIO> io(nblocks);
        for (int j = 0; j < nblocks; j++)
            io[j].start_in = -1;
        io[0].start_in = -2;
        SD_HIP(hipMemcpy(d_io.p, io.data(), io.size() * sizeof(VitBlockIO), hipMemcpyHostToDevice));
        Vit2Work vit2;
        const bool v2 = vit2_supported(vc) && !(getenv("SDHIP_VIT2") && atoi(getenv("SDHIP_VIT2")) == 0);
        if (v2)
            launch_vit_decode2(vc, (const int8_t *)d_syms, 0, nblocks, d_io.p, d_vb.p, vit2, nullptr);
        else
            launch_vit_decode(vc, (const int8_t *)d_syms, 0, nblocks, d_io.p, d_dec.p, d_vb.p, nullptr);
        SD_HIP(hipMemcpy(io.data(), d_io.p, io.size() * sizeof(VitBlockIO), hipMemcpyDeviceToHost));
        auto redo = [&](int j, int start) {
            VitBlockIO one{};
            one.start_in = start;
            SD_HIP(hipMemcpy(d_io.p + j, &one, sizeof(one), hipMemcpyHostToDevice));
            launch_vit_decode(vc, (const int8_t *)d_syms, j, 1, d_io.p + j, d_dec.p + (size_t)j * dstride, d_vb.p + (size_t)j * wpb, nullptr);
            SD_HIP(hipMemcpy(&io[j], d_io.p + j, sizeof(one), hipMemcpyDeviceToHost));
        };
        for (int j = 0; j < nblocks; j++)
        {
            if (io[j].tb_fallback == 2)
                redo(j, io[j].start_used);
            if (j > 0 && io[j].start_used != io[j - 1].ret_state)
                redo(j, io[j - 1].ret_state);
        }
        const long long nb = (long long)nblocks * frame_bits;
        hipLaunchKernelGGL(k_unpack_bits, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, nullptr, d_vb.p, wpb, frame_bits, nblocks, d_out);
        SD_HIP(hipDeviceSynchronize());
        return 0;
        SD_GUARD_END(-1)
    }

    int sdhip_op_rs_decode(int device, uint8_t *d_data, int nframes, int frame_stride, int dualbasis, int I, int rs_type, int fill_bytes, int *d_errors)
    {
        SD_GUARD_BEGIN
        SD_HIP(hipSetDevice(device));
        DevBuf<uint8_t> clean;
        clean.reserve(rs_scratch_bytes((long long)nframes * I));
        launch_rs_only(d_data, nframes, frame_stride, dualbasis, I, rs_type == SDHIP_RS239 ? 16 : 32, fill_bytes, d_errors, nullptr, clean.p);
        SD_HIP(hipDeviceSynchronize());
        return 0;
        SD_GUARD_END(-1)
    }
}
