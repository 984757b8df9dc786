/*
 * sdhip.h -- C ABI of libsdhip.so: the MI355X (gfx950) implementation of SatDump's
 * baseband -> soft symbols -> Viterbi -> deframe -> derand -> RS -> CADU hot path.
 *
 * This is the drop-in boundary. The thin C++ pipeline modules in plugin/ (subclasses of
 * the reference's satdump::pipeline::ProcessingModule, src-core/pipeline/module.h:58-191)
 * call ONLY these entry points; so does the Python binding used by tests/ and bench.py.
 * Plain pointers and sizes, no C++ / torch types, int return codes (0 = ok, <0 = error,
 * message via sdhip_last_error()), no exceptions cross this boundary.
 *
 * One handle = one stream. A handle is not thread safe; different handles may be used
 * concurrently (one per GPU / per stream). The library owns all device memory.
 *
 * Every struct field mirrors a JSON key of the reference module it replaces:
 *   sdhip_demod_cfg  <- "psk_demod"                 src-core/pipeline/modules/demod/module_psk_demod.cpp:12-84,
 *                                                   module_demod_base.cpp:12-57, module_psk_demod.h:31-39
 *   sdhip_fec_cfg    <- "ccsds_conv_concat_decoder" src-core/pipeline/modules/ccsds/module_ccsds_conv_concat_decoder.cpp:16-38
 *                       "metop_ahrpt_decoder"       plugins/noaa_metop_support/metop/module_metop_ahrpt_decoder.cpp:17-28
 */
#ifndef SDHIP_H
#define SDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

    /* ---- enums ---------------------------------------------------------------------- */
    enum
    {
        SDHIP_BPSK = 0,    /* "bpsk"    */
        SDHIP_BPSK_90 = 1, /* "bpsk_90" (decoder only) */
        SDHIP_QPSK = 2,    /* "qpsk"    */
        SDHIP_OQPSK = 3,   /* "oqpsk"   */
        SDHIP_8PSK = 4     /* "8psk" (demod only) */
    };
    enum
    {
        SDHIP_RS_NONE = 0,
        SDHIP_RS223 = 1, /* "rs223" */
        SDHIP_RS239 = 2  /* "rs239" */
    };
    enum
    {
        SDHIP_FMT_CF32 = 0, /* baseband_format "cf32" / "f32" : raw read, baseband_interface.h:172-174 */
        SDHIP_FMT_CS16 = 1, /* "cs16" / "s16" : x * (1/32767), baseband_interface.h:176-180 */
        SDHIP_FMT_CS8 = 2,  /* "cs8"  / "s8"  : x * (1/127),   baseband_interface.h:181-185 */
        SDHIP_FMT_CU8 = 3,  /* "cu8"  / "u8"  : (x - 127) * (1.0/127.0) in double, baseband_interface.h:190-198 */
        SDHIP_FMT_CS32 = 4  /* "cs32" / "s32" : x * (1/2147483647), baseband_interface.h:175-178 */
    };
    enum
    {
        SDHIP_DEC_CONV_CONCAT = 0, /* ccsds_conv_concat_decoder: Viterbi1_2 r=1/2 */
        SDHIP_DEC_METOP_AHRPT = 1, /* metop_ahrpt_decoder: Viterbi3_4 (MetOp puncture), deframer SYNCED=18, Viterbi watchdog */
        SDHIP_DEC_SIMPLE_PSK = 2,  /* ccsds_simple_psk_decoder: hard decisions (+NRZ-M / QPSK differential) -> deframer(s) -> derand -> RS
                                      (src-core/pipeline/modules/ccsds/module_ccsds_simple_psk_decoder.cpp:16-296) */
        SDHIP_DEC_FENGYUN_AHRPT = 3, /* fengyun_ahrpt_decoder: a Viterbi3_4 (fymode) per QPSK rail, FengyunDiff::work2, deframer SYNCING=8 / SYNCED=16,
                                      derand, RS223 I=4 (plugins/fengyun3_support/fengyun3/module_fengyun_ahrpt_decoder.cpp:14-126). Reads
                                      viterbi_outsync_after, viterbi_ber_thresold, invert_second_viterbi; 16384 soft bytes per read */
        SDHIP_DEC_FENGYUN_MPT = 4, /* fengyun_mpt_decoder: the same loop on two Viterbi1_2 (rate 1/2 rails, phases 0 / 90), the second rail always complemented, each
                                      rail's byte pairs exchanged in front of its decoder, the deframer's default thresholds, the Viterbi watchdog on decoder 1 alone
                                      as the module writes it (plugins/fengyun3_support/fengyun3/module_fengyun_mpt_decoder.cpp:17-134). Reads viterbi_outsync_after,
                                      viterbi_ber_thresold */
    };
    enum
    {
        SDHIP_RATE_1_2 = 0,
        SDHIP_RATE_2_3 = 1,
        SDHIP_RATE_3_4 = 2,
        SDHIP_RATE_5_6 = 3,
        SDHIP_RATE_7_8 = 4
    };

    /* ---- psk_demod ------------------------------------------------------------------- */
    typedef struct sdhip_demod_cfg
    {
        /* reference json keys */
        double samplerate;                /* "samplerate" (mandatory) */
        double symbolrate;                /* "symbolrate" */
        int constellation;                /* "constellation": SDHIP_BPSK / QPSK / OQPSK / 8PSK */
        float rrc_alpha;                  /* "rrc_alpha" (mandatory) */
        int rrc_taps;                     /* "rrc_taps", default 31 (forced odd, firdes.cpp:36) */
        float pll_bw;                     /* "pll_bw" (mandatory) */
        float agc_rate;                   /* "agc_rate", default 1e-2 */
        int dc_block;                     /* "dc_block", default 0 */
        int iq_swap;                      /* "iq_swap", default 0 */
        float min_sps, max_sps;           /* "min_sps"/"max_sps", default 1.1 / 4.0 */
        float clock_gain_omega;           /* default (8.7e-3)^2/4 */
        float clock_mu;                   /* default 0.5 */
        float clock_gain_mu;              /* default 8.7e-3 */
        float clock_omega_relative_limit; /* default 0.005 */
        float costas_max_offset_hz;       /* "costas_max_offset" in Hz; <=0 -> 1.0 rad/sample */
        int buffer_size;                  /* "buffer_size"; <=0 -> reference default (module_demod_base.cpp:22-25) */
        int post_costas_dc;               /* "post_costas_dc", default 0: a CorrectIQ DC block between the Costas loop and the clock recovery
                                             (module_psk_demod.cpp:36-38, 127-134) */
        int has_carrier;                  /* "has_carrier", default 0: BPSK on a residual carrier -- a carrier-tracking PLL and a DC block between the
                                             RRC filter and the Costas loop (module_psk_demod.cpp:39-40, 93-113; pll_carrier_tracking.cpp:23-66).
                                             BPSK only; the Costas frequency limit then defaults to 0.2 rad/sample instead of 1.0 */
        float carrier_pll_bw;             /* "carrier_pll_bw" (mandatory with has_carrier) */
        float carrier_pll_max_offset;     /* "carrier_pll_max_offset", default 3.14 rad/sample */
        /* engine knobs (ours; no reference equivalent) */
        int exact;     /* 1: one sequential lane per stream, bit-for-bit the reference schedule (slow; parity tests) */
        int chunk_len; /* speculative chunk length in (resampled) samples; <=0 -> auto */
        int warmup;    /* warm-up overlap per chunk in samples; <=0 -> auto */
        int device;    /* HIP device ordinal */
        double freq_shift; /* "freq_shift" (Hz; a long in the reference, module_demod_base.cpp:36-37): dsp::FreqShiftBlock between the DC block
                              and the resampler (module_demod_base.cpp:122-123, freq_shift.cpp:18-46). 0 = none */
        int doppler;         /* "enable_doppler" (module_demod_base.cpp:43-44, 125-171): dsp::DopplerCorrectBlock behind the frequency shift. The block's
                                target frequency comes from the pass prediction (SGP4 on the satellite's TLE, doppler_correct.cpp:68-93) once per source
                                buffer of `buffer_size` samples: host work that stays with the caller -- sdhip_demod_doppler_targets hands the targets in */
        float doppler_alpha; /* "doppler_alpha", default 0.01: the one-pole ramp of the rotator's frequency toward the target */
        double custom_samplerate; /* "custom_samplerate" (a long in the reference, module_demod_base.cpp:73-74): the sample rate the chain works at, in place of the
                                     one initb derives from min_sps / max_sps -- the resample DECISION stays initb's (input sps outside [min_sps, max_sps]). 0 = none */
    } sdhip_demod_cfg;

    typedef struct sdhip_demod_stats
    {
        uint64_t samples_in;   /* input samples consumed */
        uint64_t symbols_out;  /* symbols produced */
        float freq_hz;         /* Costas frequency, rad_to_hz(freq, final_samplerate) ("freq" stat) */
        float final_sps;       /* samples per symbol after the resample decision */
        float final_samplerate;
        int buffer_size;       /* effective d_buffer_size */
        int resample_interp, resample_decim; /* 0,0 when not resampling */
        uint32_t chunks;        /* speculative chunks processed by the last call (all three loop stages) */
        uint32_t chunks_fixed;  /* chunks re-run from the exact boundary state because the certificate failed */
        uint32_t chunks_rotated;/* Costas chunks that locked on another constellation symmetry and were rotated back */
        uint32_t chunks_inexact;/* chunks accepted by tolerance (boundary states equal to ~1e-6) rather than bit-for-bit */
        uint32_t chunks_forced; /* boundaries let through after the re-run round limit: the signal was not locked there (noise) */
    } sdhip_demod_stats;

    void sdhip_demod_cfg_default(sdhip_demod_cfg *cfg);
    /* returns NULL on error */
    void *sdhip_demod_create(const sdhip_demod_cfg *cfg);
    void sdhip_demod_destroy(void *h);
    /* Host-buffer path: append nsamples complex samples in format fmt (SDHIP_FMT_*). Processing is
       deferred until enough samples are pending or sdhip_demod_flush() is called. */
    int sdhip_demod_push(void *h, const void *iq, size_t nsamples, int fmt);
    /* Process everything pending, including the final partial chunk. */
    int sdhip_demod_flush(void *h);
    /* Pop up to cap soft-symbol BYTES (int8; BPSK 1 B/symbol = I*50, others 2 B/symbol = I*100,Q*100,
       module_psk_demod.cpp:199-213). Returns the number of bytes written, <0 on error. */
    int64_t sdhip_demod_pull(void *h, int8_t *soft, size_t cap);
    /* Device-resident path: d_iq points to nsamples complex samples ALREADY IN HBM on cfg->device;
       soft symbols are written to d_soft (device, capacity soft_cap bytes). If d_syms != NULL the float
       symbols (2 floats each) are also written (capacity syms_cap symbols). `final` != 0 also drains the
       tail. Returns soft bytes written, <0 on error. Stream state carries across calls. */
    int64_t sdhip_demod_process_dev(void *h, const void *d_iq, size_t nsamples, int fmt, int8_t *d_soft, size_t soft_cap, float *d_syms, size_t syms_cap, int final);
    int sdhip_demod_get_stats(void *h, sdhip_demod_stats *st);
    /* TEST TAP (tests/test_demod_gpu.py::test_every_symbol_beyond_tolerance_is_an_arm_flip; not a product feature). mode 1: from the next call on, the chunk-parallel
       mode's d_syms output carries, in the eight bytes of every symbol, an int64 -- the clock recovery's position on its interpolator grid for that symbol,
       (input sample index of the call * 128 + arm), arm = the index rint(mu * 128) of clock_recovery_mm.cpp:66 -- instead of the symbol (the int8 output is then
       meaningless). Two fresh handles on the same input follow the same trajectory, so one run gives the symbols and a second one their grid positions. mode 0: off. */
    int sdhip_demod_set_tap(void *h, int mode);
    /* Doppler correction (cfg.doppler): append the rotator's target frequencies, rad / sample, for the source buffers to come. targets[k] is what
       DopplerCorrectBlock::work computes BEHIND a buffer -- hz_to_rad(-doppler_shift, samplerate) at the time that buffer ends (doppler_correct.cpp:68-93) --
       and is in force during the next one; the first buffer of a stream runs on target 0, as in the reference. A buffer is `buffer_size` samples
       (sdhip_demod_get_stats tells the effective value); processing fails when a buffer starts for which no target has been handed in. */
    int sdhip_demod_doppler_targets(void *h, const float *targets, size_t n);

    /* The DVB-S2 demodulator's front end (plugins/dvb_support/dvbs2/module_dvbs2_demod.cpp:98-105 on BaseDemodModule: resampler, AGC, RRC filter,
       M&M clock recovery -- psk_demod's stages without its Costas loop; carrier recovery follows per frame, sdhip_s2_pl_sync_dev /
       sdhip_s2_pll_dev). Same configuration struct (constellation only selects the clock recovery's OQPSK handling: pass SDHIP_QPSK; pll_bw
       is unused), same handle functions: sdhip_demod_process_dev with d_syms != NULL delivers the clock-recovered symbols (the int8 soft
       output is a by-product). Destroy with sdhip_demod_destroy. */
    void *sdhip_dvbs2_front_create(const sdhip_demod_cfg *cfg);

    /* ---- ndsp: the reference's new block API -------------------------
       satdump::ndsp::PSKDemodHierBlock (src-core/dsp/hier/psk_demod.h:22-249, psk_demod.cpp:8-14): RRC FIR -> AGC (reference 0.6) ->
       M&M clock recovery -> Costas loop at ONE sample per symbol, complex symbols out; no resampler, no quantiser. The fields are the
       block's set_cfg() keys; the defaults (sdhip_ndsp_psk_cfg_default) are the member blocks' own:
       rrc.h:17-21, agc.h:14-17, clock_recovery_mm.h:17-23, costas.h:14-16. */
    typedef struct sdhip_ndsp_psk_cfg
    {
        int device;
        int constellation;      /* "constellation": SDHIP_BPSK / SDHIP_QPSK (psk_demod.h:205-214; the block logs TODOREWORK for oqpsk) */
        double samplerate;      /* "samplerate" (default 6e6) */
        double symbolrate;      /* "symbolrate" (default 2e6) */
        double rrc_gain;        /* "rrc_gain" 1 */
        double rrc_alpha;       /* "rrc_alpha" 0.35 */
        int rrc_ntaps;          /* "rrc_ntaps" 31 */
        float agc_rate;         /* "agc_rate" 1e-4 */
        float agc_reference;    /* "agc_reference" 0.6 (set by the hier block's constructor) */
        float agc_gain;         /* "agc_gain" 1 */
        float agc_max_gain;     /* "agc_max_gain" 65536 */
        float rec_omega;        /* "rec_omega": 0 = samplerate / symbolrate (what setting either rate does, psk_demod.h:224) */
        float rec_omegaGain;    /* "rec_omegaGain" pow(8.7e-3, 2) / 4 */
        float rec_mu;           /* "rec_mu" 0.5 */
        float rec_muGain;       /* "rec_muGain" 8.7e-3 */
        float rec_omegaLimit;   /* "rec_omegaLimit" 0.005 */
        int rec_nfilt;          /* "rec_nfilt" 128 (the only bank shape the HIP path carries; anything else is refused) */
        int rec_ntaps;          /* "rec_ntaps" 8 (likewise) */
        float pll_loop_bw;      /* "pll_loop_bw" 0.004 */
        float pll_freq_limit;   /* "pll_freq_limit" 1.0 */
        int exact;              /* 1 = one sequential lane per loop, the reference's float operations in its order (bit-exact symbols) */
        int chunk_len;          /* 0 = automatic */
        int warmup;             /* 0 = automatic */
    } sdhip_ndsp_psk_cfg;
    void sdhip_ndsp_psk_cfg_default(sdhip_ndsp_psk_cfg *c);
    void *sdhip_ndsp_psk_demod_create(const sdhip_ndsp_psk_cfg *cfg);
    /* ONE member block of that chain as a handle of its own -- the flowgraph registry's single nodes (dsp/flowgraph/dsp_flowgraph_register.cpp): the same
       struct carries the block's keys (rrc_* + samplerate / symbolrate for the filter design; agc_*; rec_* with rec_omega or samplerate / symbolrate;
       pll_* + constellation for the loop's order), the same work / stats / destroy functions serve it. Each is the stage the hier block runs, with its
       state carried across work() calls, exact = 1 bit for bit the reference block (dsp/filter/fir.cpp:62-133 incl. its ntaps-sample latency,
       dsp/agc/agc.cpp:22-39, dsp/clock_recovery/clock_recovery_mm.cpp:66-183, dsp/pll/costas.cpp:12-61). */
    enum
    {
        SDHIP_NDSP_HIER = 0,    /* "psk_demod_cc" */
        SDHIP_NDSP_RRC_FIR = 1, /* "rrc_fir_cc" */
        SDHIP_NDSP_AGC = 2,     /* "agc_cc" */
        SDHIP_NDSP_MM = 3,      /* "clock_recovery_mm_cc" */
        SDHIP_NDSP_COSTAS = 4,  /* "costas_cc" */
        SDHIP_NDSP_GARDNER = 5, /* "clock_recovery_gardner_cc" (dsp/clock_recovery/clock_recovery_gardner.cpp): rec_* keys as for the M&M block */
        SDHIP_NDSP_AGC_FAST = 6, /* "agc_fast_cc" (dsp/agc/agc_fast.cpp:22-58, dsp_flowgraph_register.cpp:278): the gain follows |input| x gain -- the magnitudes of the INPUT
                                   taken first (volk_32fc_magnitude_32f) -- instead of |output|; agc_* keys as for the AGC block */
        SDHIP_NDSP_COSTAS_FAST = 7, /* "costas_fast_cc" (dsp/pll/costas_fast.cpp:15-110, dsp_flowgraph_register.cpp:294): the VCO as a complex number turned by small-angle
                                   updates, renormalised every 65th sample; pll_* keys + constellation for the order as for the Costas block. exact = 1: one sequential lane.
                                   exact = 0: lane-per-chunk with a STRICT hand-off -- a chunk stands only if its start state is bit-identical to its predecessor's end
                                   state modulo an exact quarter / half turn, else it runs again from that state; a call whose lanes do not hand off, or whose frequency
                                   limiter comes into play, runs as the one sequential lane -- so the samples are the block's own, float for float, in either mode
                                   (the "freq" statistic within ~1e-6 rad / sample). SDHIP_CF_STRICT=0: the plain loop's tolerance windows instead */
        SDHIP_NDSP_MM_FAST = 8 /* "fast_clock_recovery_mm_cc" (dsp/clock_recovery/clock_recovery_mm_fast.cpp:66-163, dsp_flowgraph_register.cpp:306): the M&M detector on a linear
                                   interpolation, the rate term updated every fifth symbol; rec_omega / rec_omegaGain / rec_mu / rec_muGain / rec_omegaLimit. exact = 1: one
                                   sequential lane. exact = 0: a lane per (chunk, value of the every-fifth-symbol counter at the warm-up's start) -- how many symbols lie in
                                   front of a chunk nobody knows ahead, so each chunk runs under all five cadences -- and a STRICT hand-off: the variant stands whose state at
                                   the chunk start is bit for bit its predecessor's state at its end, a chunk with none runs again from that state; a call whose lanes do not
                                   hand off runs as the one sequential lane. The symbols are the block's own, float for float, in either mode */
    };
    void *sdhip_ndsp_block_create(int kind, const sdhip_ndsp_psk_cfg *cfg);
    void sdhip_ndsp_psk_demod_destroy(void *h);
    /* One DSPBuffer's worth of work() of the whole hier block: nsamples complex floats (device) in, the symbols it produces (complex
       floats, device, capacity out_cap symbols) out. Returns the symbols written, <0 on error. The stream state (filter history with the
       FIR block's ntaps-sample latency, gain, clock and loop state) carries across calls, so the output does not depend on how the
       stream is cut into buffers -- as in the reference. */
    int64_t sdhip_ndsp_psk_demod_work_dev(void *h, const float *d_in, size_t nsamples, float *d_out, size_t out_cap);
    /* the same through host buffers */
    int64_t sdhip_ndsp_psk_demod_work(void *h, const float *in, size_t nsamples, float *out, size_t out_cap);
    /* get_cfg("pll_freq") (rad_to_hz(freq, symbolrate), psk_demod.h:170) and the chunk statistics of the last call */
    int sdhip_ndsp_psk_demod_get_stats(void *h, sdhip_demod_stats *st);

    /* ---- ccsds_conv_concat_decoder / metop_ahrpt_decoder ------------------------------ */
    typedef struct sdhip_fec_cfg
    {
        int decoder;               /* SDHIP_DEC_* */
        int constellation;         /* "constellation": bpsk / bpsk_90 / qpsk / oqpsk */
        int iq_invert;             /* "iq_invert" */
        int cadu_size;             /* "cadu_size" in BITS incl. ASM */
        int viterbi_outsync_after; /* "viterbi_outsync_after" */
        float viterbi_ber_thresold;/* "viterbi_ber_thresold" (sic) */
        int nrzm;                  /* "nrzm" */
        int derandomize;           /* "derandomize", default 1 */
        int derand_after_rs;       /* "derand_after_rs", default 0 */
        int derand_start;          /* "derand_start", default 4 */
        int rs_i;                  /* "rs_i"; 0 disables RS */
        int rs_fill_bytes;         /* "rs_fill_bytes", default -1 */
        int rs_dualbasis;          /* "rs_dualbasis", default 1 */
        int rs_type;               /* SDHIP_RS_* ("rs_type") */
        int rs_usecheck;           /* "rs_usecheck" */
        uint32_t asm_sync;         /* "asm", default 0x1ACFFC1D */
        /* ccsds_simple_psk_decoder only (module_ccsds_simple_psk_decoder.cpp:25-29) */
        int qpsk_swap_iq;          /* "qpsk_swap_iq", default 0 */
        int qpsk_swap_diff;        /* "qpsk_swap_diff", default 1 */
        int oqpsk_delay;           /* "oqpsk_delay", default 0 */
        int oqpsk_method2;         /* "oqpsk_method2", default 0 */
        int oqpsk_method3;         /* "oqpsk_method3", default 0 */
        /* ccsds_conv_concat_decoder only: "conv_rate" (module_ccsds_conv_concat_decoder.cpp:33,93-119). 0 = "1/2" (Viterbi1_2);
           SDHIP_RATE_2_3 .. SDHIP_RATE_7_8 = the punctured rates of viterbi::Viterbi_Depunc (viterbi_punc.cpp, depunc.h) */
        int conv_rate;
        /* engine knobs */
        int device;
        /* fengyun_ahrpt_decoder only: "invert_second_viterbi" (module_fengyun_ahrpt_decoder.cpp:17,67) */
        int invert_second_viterbi;
        /* meteor_lrpt_decoder, "m2x_mode" + "interleaved" (module_meteor_lrpt_decoder.cpp:28-41,103-199): with decoder = SDHIP_DEC_CONV_CONCAT and constellation =
           SDHIP_OQPSK (the module's Viterbi1_2 over {PHASE_0, PHASE_90} with the I/Q exchange searched), the input is the INTERLEAVED .soft stream: two
           meteor::DeinterleaverReader (the stream, and the stream a quarter turn on) in front of two such Viterbis, per read the locked one's bits to the deframer.
           Held to the module's loop with its sample reader put right (in the reference tree it reports an error after 8192 bytes and the branch decodes nothing:
           oracle/ref_wrap_lrpt_m2x.cpp, tests/test_lrpt_m2x_reference_cpu.py). Input of any length per call; sdhip_fec_flush ends the stream. */
        int m2x_interleaved;
    } sdhip_fec_cfg;

    typedef struct sdhip_fec_stats
    {
        uint64_t soft_in;        /* soft bytes consumed */
        uint64_t blocks;         /* Viterbi blocks processed */
        uint64_t bits_decoded;   /* Viterbi output bits fed to the deframer */
        uint64_t frames_deframed;/* frames emitted by the deframer */
        uint64_t frames_out;     /* frames written (after rs_usecheck) */
        float viterbi_ber;       /* "viterbi_ber" */
        int viterbi_lock;        /* "viterbi_lock": 0 NOSYNC, 1 SYNCED */
        int deframer_state;      /* numeric threshold state: 2 NOSYNC, 6 SYNCING, 12/18 SYNCED */
        int rs_errors[8];        /* last frame's per-codeword error counts (-1 = uncorrectable) */
        uint32_t vit_respec;     /* Viterbi blocks re-decoded because the start-state speculation failed */
        uint32_t tb_respec;      /* traceback segments re-run because the merge certificate failed */
        float viterbi2_ber;      /* fengyun_ahrpt_decoder: "viterbi2_ber" / "viterbi2_lock" (viterbi_ber / viterbi_lock are its viterbi1_*) */
        int viterbi2_lock;
        /* watchdog actions of the decoder MODULES since the handle was created: MetOp's viterbi.reset() after 10 NOSYNC reads (module_metop_ahrpt_decoder.cpp:58-66),
           the FengYun modules' `shift` / `invert_branches` toggles AND every read their cumulative viterbiNoSyncRun counted (module_fengyun_ahrpt_decoder.cpp:82-114).
           A sharded decode (hip_devices) is only the single stream's when this stays put over every shard's own run: a shard starts those counters cold. */
        uint32_t watchdog_events;
    } sdhip_fec_stats;

    void sdhip_fec_cfg_default(sdhip_fec_cfg *cfg);
    void *sdhip_fec_create(const sdhip_fec_cfg *cfg);
    void sdhip_fec_destroy(void *h);
    /* Host-buffer path: append n soft bytes (the .soft wire format). Whole Viterbi blocks
       (max(cadu_size,8192) bytes, 16384 for MetOp) are decoded; the remainder stays pending. */
    int sdhip_fec_push(void *h, const int8_t *soft, size_t n);
    /* Pop up to cap_frames CADUs (ceil(cadu_size/8) bytes each) into cadu. Returns frame count. */
    int64_t sdhip_fec_pull(void *h, uint8_t *cadu, size_t cap_frames);
    /* Device-resident path: d_soft holds n soft bytes in HBM; CADUs are written to d_cadu (device,
       capacity cap_frames frames). Returns frames written, <0 on error. */
    int64_t sdhip_fec_process_dev(void *h, const int8_t *d_soft, size_t n, uint8_t *d_cadu, size_t cap_frames);
    /* End of the input (m2x_interleaved only; a no-op otherwise): the reads the module's loop still makes on the zero-filled rest of its FIFOs until should_run()
       turns false. d_cadu / cap_frames as for sdhip_fec_process_dev, or NULL / 0: the frames queue up for sdhip_fec_pull. Returns the frames written / queued. */
    int64_t sdhip_fec_flush(void *h, uint8_t *d_cadu, size_t cap_frames);
    int sdhip_fec_get_stats(void *h, sdhip_fec_stats *st);
    /* Optional per-block taps of the last process call (host arrays, may be NULL):
       blk_ber[nblocks], blk_state[nblocks]. Returns number of blocks. (fengyun_ahrpt_decoder: two entries per read, Viterbi 1 then Viterbi 2.) */
    int64_t sdhip_fec_get_block_taps(void *h, float *blk_ber, int *blk_state, size_t cap);

    /* ---- kernel-level entry points (unit parity tests; each replaces one reference function) ---- */
    /* viterbi::CCDecoder::work chained over nblocks (cc_decoder.cpp:295-302). d_syms: per block
       2*(frame_bits+6) unsigned soft symbols (device). d_out: frame_bits bytes/block, one bit per byte. */
    int sdhip_op_ccdecoder(int device, int frame_bits, const uint8_t *d_syms, int nblocks, uint8_t *d_out);
    /* viterbi::Viterbi27::work over nframes consecutive calls of one decoder (src-core/common/codings/viterbi/viterbi27.cpp:31-66; the
       decoder of the Meteor LRPT / Inmarsat plugin modules): d_soft = nframes x 2*frame_bits int8 soft symbols, d_out = nframes x
       frame_bits/8 bytes, ber_out (host, may be NULL) = Viterbi27::ber() after each call. CCSDS polys {79, 109} only; soft input only. */
    int sdhip_op_viterbi27(int device, int frame_bits, int ber_test_size, const int8_t *d_soft, int nframes, uint8_t *d_out, float *ber_out);
    /* reedsolomon::ReedSolomon::decode_interlaved over nframes (reedsolomon.cpp:53-116). d_data points at
       the first codeblock byte of frame 0 (cadu+4); errors: nframes*I ints (device). fill_bytes as in the reference. */
    int sdhip_op_rs_decode(int device, uint8_t *d_data, int nframes, int frame_stride, int dualbasis, int I, int rs_type, int fill_bytes, int *d_errors);
    /* dsp blocks over one stream, EXACT sequential semantics (single lane), for arithmetic parity:
       kind 0 AGC(rate,ref,gain,max) agc.cpp:25-39 | 1 RRC FIR(fs,symrate,alpha,ntaps) fir.cpp:74-83 |
       2 Costas(bw,order,limit) costas_loop.cpp:23-65 | 3 MM(omega,gw,mu,gmu,lim) clock_recovery_mm.cpp:52-121 |
       4 rational resampler(interp,decim) rational_resampler.cpp:43-64 | 5 DC block correct_iq.cpp:18-35 |
       7 Gardner(omega,gw,mu,gmu,lim) clock_recovery_gardner.cpp:33-124 | 8 carrier PLL(bw,max,min) pll_carrier_tracking.cpp:8-66 |
       9 ndsp Costas(bw,order,limit) dsp/pll/costas.cpp:12-61 (branched clip) | 10 ndsp Gardner(omega,gw,mu,gmu,lim)
       dsp/clock_recovery/clock_recovery_gardner.cpp:60-170 (kind 7 with branched clips on floats). The other ndsp blocks compute what kinds 0, 1 and 3 do
       (dsp/agc/agc.cpp:22-39, dsp/filter/fir.cpp:62-133 minus its ntaps-sample latency, dsp/clock_recovery/clock_recovery_mm.cpp:66-183).
       Returns output sample count. */
    int64_t sdhip_op_block(int device, int kind, const float *params, const float *d_in, size_t n, float *d_out, size_t out_cap);

    /* The coefficient tables the modules' blocks are constructed with, designed on the host exactly as the reference designs them
       (double precision, float taps): kind 0 = dsp::firdes::root_raised_cosine(gain, fs, symrate, alpha, ntaps) firdes.cpp:34-78,
       params[5]; kind 1 = the M&M interpolator bank, windowed_sinc + nuttall split over nfilt arms (clock_recovery_mm.cpp:22-24,
       window.cpp:9-50, polyphase_bank.cpp:6-39), params {nfilt, ntaps}, dims = {nfilt, taps per arm}; kind 2 = the rational
       resampler's bank (design_resampler_filter_float, firdes.cpp:276-301), params {interp, decim}, dims = {interp and decim
       reduced by their gcd, taps per arm}. No device is touched. Returns the number of floats written, <0 on error. */
    int64_t sdhip_design(int kind, const double *params, float *out, size_t cap, int *dims);

    /* ---- measurement ------------------------------------------------------------------ */
    /* Per-kernel timing with HIP events recorded on the launch stream around every kernel launch of the
       library (process-wide; off by default). sdhip_prof_get(idx, ...) returns the number of distinct kernels
       seen and, for 0 <= idx < that number, the kernel's name, total milliseconds and launch count. */
    void sdhip_prof_enable(int on);
    void sdhip_prof_reset(void);
    int sdhip_prof_get(int idx, char *name, size_t name_cap, double *total_ms, long long *launches);

    /* ---- memory ----------------------------------------------------------------------- */
    /* Park the device / pinned blocks of destroyed handles (per device and size) and hand them to later handles instead of
       returning them to the driver: a caller that starts every recording (or every chunk of a sharded recording) with fresh
       handles -- the reference constructs new module instances per pipeline run, src-core/pipeline/pipeline_run.cpp:40-70 -- then
       allocates once. Off by default; sdhip_pool_enable(0) and sdhip_pool_trim() release what is parked. Process-wide. */
    void sdhip_pool_enable(int on);
    void sdhip_pool_trim(void);

    /* ---- DVB-S2 LDPC soft decoder (BASELINE configs[4]) ---------------------------------------------------
       Replaces dvbs2::BBFrameLDPC (plugins/dvb_support/codings/dvb-s2/bbframe_ldpc.{h,cpp}; decoder = ldpc/layered_decoder.hh with
       OffsetMinSumAlgorithm<SIMD<int8_t, W>, NormalUpdate, 2>, ldpc/algorithms.hh:207-279) as DVBS2DemodModule::process_s2 calls it
       (plugins/dvb_support/dvbs2/module_dvbs2_demod.cpp:246-257): int8 soft bits in (positive = bit 0), the same soft bits updated in
       place out, bit for bit the reference's. */
    typedef struct sdhip_ldpc_cfg
    {
        int framesize; /* dvbs2_framesize_t: 0 FECFRAME_NORMAL (64800), 1 FECFRAME_SHORT (16200) */
        int rate;      /* dvbs2_code_rate_t: 0 C1_4, 1 C1_3, 2 C2_5, 3 C1_2, 4 C3_5, 5 C2_3, 6 C3_4, 7 C4_5, 8 C5_6, 9 C7_8 (no LDPC table: error),
                          10 C8_9, 11 C9_10 (common/codings/dvb-s2/dvbs2.h:9-23) */
        int batch;     /* dvbs2::simd_type::SIZE of the reference build being replaced: frames per BBFrameLDPC::decode call that share ONE
                          early exit (16 with -msse4.1 -- plugins/dvb_support/CMakeLists.txt:25-38 -- 1 for the generic build) */
        int device;
    } sdhip_ldpc_cfg;
    typedef struct sdhip_ldpc_info
    {
        int code_len, data_len;      /* LDPCInterface::code_len / data_len (BBFrameLDPC::dataSize) */
        int layers;                  /* q = (N - K) / 360 */
        int links_total;             /* edges of the Tanner graph */
        int max_phases;              /* > 1: some layer holds checks that share a data bit and is run in that many dependent steps */
        int layers_with_shared_bits;
        uint64_t msg_bytes_per_frame; /* check-to-bit message state kept in HBM per frame */
    } sdhip_ldpc_info;
    void *sdhip_ldpc_create(const sdhip_ldpc_cfg *cfg); /* NULL on error (unknown rate / size: the reference's ctor would leave ldpc unset) */
    void sdhip_ldpc_destroy(void *h);
    int sdhip_ldpc_get_info(void *h, sdhip_ldpc_info *out);
    /* nframes (a multiple of batch) frames of code_len int8 each, consecutive, decoded in place; trials[b] for batch b = what
       BBFrameLDPC::decode returns for that call: the update passes it ran, or -1 if the batch did not converge within max_trials
       (bbframe_ldpc.cpp:114-124). Returns the number of trial launches - 1 (>= 0), < 0 on error. _dev: pointers on cfg->device, d_frames 4-byte
       aligned (refused otherwise: the kernel moves the frames as 32-bit words). */
    int sdhip_ldpc_decode_dev(void *h, int8_t *d_frames, int nframes, int max_trials, int *d_trials);
    int sdhip_ldpc_decode(void *h, int8_t *frames, int nframes, int max_trials, int *trials);

    /* ---- DVB-S2 BCH outer decoder + hard-decision repack (same path) --------------------------------------------------------------
       Replaces dvbs2::BBFrameBCH::decode (plugins/dvb_support/codings/dvb-s2/bbframe_bch.cpp:412-440 -> bch/
       bose_chaudhuri_hocquenghem_decoder.hh, bch/reed_solomon_error_correction.hh) and the repack in front of it
       (plugins/dvb_support/dvbs2/module_dvbs2_demod.cpp:262-266). */
    typedef struct sdhip_bch_cfg
    {
        int framesize; /* as sdhip_ldpc_cfg */
        int rate;
        int device;
    } sdhip_bch_cfg;
    void *sdhip_bch_create(const sdhip_bch_cfg *cfg);
    void sdhip_bch_destroy(void *h);
    int sdhip_bch_dims(void *h, int *kbch, int *nbch); /* BBFrameBCH::dataSize() and the frame length (= the LDPC code's data_len) */
    /* nframes packed frames (nbch / 8 bytes each, `stride` bytes apart), corrected in place; corrections[f] = BBFrameBCH::decode's return
       value: bits corrected, 0 for a clean frame, -1 when the decoder gives up. _dev: pointers on cfg->device. */
    int sdhip_bch_decode_dev(void *h, uint8_t *d_frames, int nframes, int stride, int *d_corrections);
    int sdhip_bch_decode(void *h, uint8_t *frames, int nframes, int stride, int *corrections);
        /* dvbs2::S2BBToSoft::work (plugins/dvb_support/dvbs2/dvbs2_bb_to_soft.cpp:18-69) on nframes PL-synchronised, phase-recovered PLFRAMEs as
       S2PLLBlock hands them over (dvbs2_pll.cpp:31-49): d_plframes = nframes x frame_stride complex floats (device), each [90 header symbols |
       frame_slot_count x 90 symbols ...]. modcod / shortframes / pilots: the module's parameters (module_dvbs2_demod.cpp:55-63; the MODCOD
       table of codings/dvb-s2/modcod_to_cfg.h:19-151 decides modulation, slots and code rate; 32APSK -- no demapper table in the reference --
       and unknown MODCODs are refused with the reference's messages). lut_bits (HOST): the soft-demapper table constellation_t::make_lut
       (lut_resolution) built, [x][y][bit] int8 (module_dvbs2_demod.cpp:123-124: resolution 256) -- data the caller owns, cached on the
       device until it changes. Per frame: PLS decode (d_pls[f] = MODCOD << 2 | SHORTFRAMES << 1 | PILOTS as decoded from the header,
       dvbs2_bb_to_soft.cpp:27-51; may be NULL), PL descrambling, table lookup, the pilots branch as the block has it, de-interleaver.
       d_soft: nframes x 64800 (16200) soft bits, the LDPC decoder's input. Returns the soft bits per frame, <0 on error. */
    int sdhip_s2_bb_to_soft_dev(int device, int modcod, int shortframes, int pilots, const float *d_plframes, int frame_stride, int nframes, const int8_t *lut_bits,
                                int lut_resolution, int8_t *d_soft, int *d_pls);
    /* dvbs2::S2PLSyncBlock::work2 (plugins/dvb_support/dvbs2/dvbs2_pl_sync.cpp:52-125) over a batch of clock-recovered symbols (device, complex
       floats): the block's ring buffer is [d_syms, d_syms + nsyms). Per frame: the differential SOF + PLS correlation at every offset of a
       raw_frame_size window ((slot_number + 1) * 90 symbols, + 36 per pilot block as the constructor counts them), the first offset above
       `thresold` (the block's public member, 0.6) or else the best one, re-alignment by that offset; d_frames gets raw_frame_size symbols per
       frame at frame_stride complex floats. Frames are emitted while the input holds a frame's window plus its re-alignment symbols;
       *consumed = symbols taken out of the ring (the caller keeps the rest in front of the next call's symbols, as the ring buffer does);
       best_pos_out (HOST, may be NULL): the offset each frame was found at (0 in lock). Returns the frames written, <0 on error. */
    int64_t sdhip_s2_pl_sync_dev(int device, int slot_number, int pilots, float thresold, const float *d_syms, size_t nsyms, float *d_frames, int frame_stride,
                                 size_t max_frames, size_t *consumed, int *best_pos_out);
    /* dvbs2::S2PLLBlock::work (plugins/dvb_support/dvbs2/dvbs2_pll.cpp:20-66) over nframes synchronised frames (device, frame_stride complex floats
       each, as sdhip_s2_pl_sync_dev writes them): the frame PLL, one sequential loop over every symbol with its state carried across frames and
       calls -- state2 (HOST, in / out) = {phase, freq}, both 0 for a new stream. Header symbols: phase error against the known SOF / PLS
       symbols of the CONFIGURED modcod / shortframes / pilots (pls_code, module_dvbs2_demod.cpp:117), output as the block's "45 degree BPSK";
       behind the header: the demapper table's phase_error entries -- lut_phase_error (HOST): constellation_t::make_lut(lut_resolution)'s
       [x][y].phase_error floats, the caller's data like sdhip_s2_bb_to_soft_dev's bits. With pilots the block walks (slots + 1) * 90 + 36
       symbols (it counts one pilot block, dvbs2_pll.h:33-47) and leaves the rest of the frame unwritten: d_frames_out gets exactly what the
       block writes. Exact arithmetic only (glibc's sinf / cosf / atan2f restated), one lane: the loop is a serial chain. Returns the symbols
       walked per frame, <0 on error. */
    int sdhip_s2_pll_dev(int device, int modcod, int shortframes, int pilots, float loop_bw, const float *d_frames_in, float *d_frames_out, int frame_stride, int nframes,
                         const float *lut_phase_error, int lut_resolution, float *state2);
    /* The same loop with its FRAME-PARALLEL schedule (round 4). mode 1 = sdhip_s2_pll_dev (the serial lane, exact). mode 0 / 2: the chain of loop steps
       over the batch is cut into lanes; a lane starts a warm-up in front of its range from a data-aided estimate (phase from the 90 known header
       symbols of its frame, frequency from two consecutive headers -- the carried loop frequency only picks the 2 pi / frame branch), and the
       chain is certified lane by lane against the predecessors' end states; lanes that miss are re-run from the exact predecessor state. mode 2 =
       a new stream: the first frames (65 536 symbols) are walked by the serial lane until the loop frequency is there; mode 0 = the caller vouches
       that state2 is a locked loop's. What it promises: NOT the serial loop's symbols to 1e-5 -- the loop's detector is a 256 x 256 table, two
       trajectories on the same symbols stay ~1e-2 (8PSK, 10 dB) ... 2e-4 (QPSK, 8 dB) of a symbol apart for good -- but the decoders' output, the
       same BBFRAMEs. stats4 (HOST, may be NULL) = {lanes, lanes re-run, boundaries forced after the round limit, frames walked serially}. */
    int sdhip_s2_pll_frames_dev(int device, int modcod, int shortframes, int pilots, float loop_bw, const float *d_frames_in, float *d_frames_out, int frame_stride, int nframes,
                                const float *lut_phase_error, int lut_resolution, float *state2, int mode, unsigned *stats4);
    /* unit entry: d_out[i] = atan2f(d_y[i], d_x[i]) as the frame PLL evaluates it (glibc 2.35's float code restated) */
    int sdhip_op_atan2f(int device, const float *d_y, const float *d_x, int n, float *d_out);
    /* get_dvbs2_cfg's answer for a MODCOD: bits per symbol, slots per frame, dvbs2_code_rate_t, dvbs2_constellation_t */
    /* ---- dvbs2_ts_extractor (plugins/dvb_support/dvbs2/module_s2_ts_extractor.cpp:77-105 -> dvbs2::BBFrameTSParser::work, src-core/common/codings/dvb-s2/
       bbframe_ts_parser.cpp:96-243): MPEG-TS packets (188 bytes) out of the BBFRAMEs' data fields, one frame per parser call, the parser's state carried from
       frame to frame and from call to call. bbframe_bits = BBFrameBCH::dataSize() (or the module's "bb_size"); frames are bbframe_bits / 8 bytes apart.
       Returns the packets written (< 0: error). The _dev entry takes and leaves everything in HBM (the BBFRAMEs sdhip_dvbs2_demod_* produced). */
    typedef struct sdhip_s2_ts_stats
    {
        uint64_t frames_in, packets_out, header_crc_fails, resyncs;
        int synched;
    } sdhip_s2_ts_stats;
    void *sdhip_s2_ts_create(int device, int bbframe_bits);
    void sdhip_s2_ts_destroy(void *h);
    int64_t sdhip_s2_ts_process_dev(void *h, const uint8_t *d_bbframes, int nframes, uint8_t *d_ts, size_t cap_packets);
    int64_t sdhip_s2_ts_process(void *h, const uint8_t *bbframes, int nframes, uint8_t *ts, size_t cap_packets);
    int sdhip_s2_ts_get_stats(void *h, sdhip_s2_ts_stats *st);

    int sdhip_s2_cfg(int modcod, int shortframes, int *bits, int *slots, int *rate, int *constellation);
    /* dvbs2::S2Deinterleaver::deinterleave (codings/dvb-s2/s2_deinterleaver.cpp:92-145) over nframes frames of 64800 / 16200 soft bits:
       constellation = dvbs2_constellation_t (0 QPSK, 1 8PSK, 2 16APSK, 3 32APSK), d_in != d_out (device pointers) */
    int sdhip_s2_deinterleave_dev(int device, int constellation, int framesize, int rate, const int8_t *d_in, int8_t *d_out, int nframes);
    /* dvbs2::BBFrameDescrambler::work (bbframe_descramble.cpp:133-139) on the first kbch / 8 bytes of every frame, in place (device) */
    int sdhip_bb_descramble_dev(void *h, uint8_t *d_frames, int nframes, int stride);
    /* bit i of a frame = (soft[i] < 0), MSB first, for the first nbch soft bits of every LDPC frame (8-byte aligned, soft_stride apart) */
    int sdhip_s2_pack_dev(void *h, const int8_t *d_soft, int soft_stride, int nframes, uint8_t *d_out, int out_stride);

    /* ---- the DVB-S2 demodulator as a module-shaped handle (BASELINE configs[4]) -------------------------------------------------------------------
       Replaces satdump::pipeline::dvb::DVBS2DemodModule (plugins/dvb_support/dvbs2/module_dvbs2_demod.{h,cpp}: constructor :13-83, init :85-137,
       process :141-222, process_s2 :239-293, getModuleStats :224-237): baseband samples in, BBFRAME bytes out (bch_decoder->dataSize() / 8 bytes per
       frame: what the module writes to its .bbframe file / output fifo). The fields are the module's JSON keys; `front` carries BaseDemodModule's and
       the RRC / clock recovery keys (its pll_bw = the module's "pll_bw", the FRAME PLL's loop bandwidth; constellation is ignored; exact = 1 selects the
       serial schedules everywhere: bit for bit the reference's blocks chained the way the module chains them, freq_prop_factor = 0 only). */
    typedef struct sdhip_dvbs2_cfg
    {
        sdhip_demod_cfg front;        /* "samplerate", "symbolrate", "rrc_alpha", "rrc_taps", "pll_bw", "agc_rate", "dc_block", "iq_swap", "buffer_size", "clock_*" ... */
        float freq_prop_factor;       /* "freq_prop_factor", default 0.01: share of the PLL's frequency handed to the rotator in front of the PL
                                         synchroniser per frame (module_dvbs2_demod.cpp:204-206). Applied at call boundaries in closed form -- see
                                         dvbs2_engine.hip's header: the reference's own feedback runs on thread timing */
        int modcod;                   /* "modcod" (mandatory) */
        int shortframes;              /* "shortframes" */
        int pilots;                   /* "pilots" */
        float sof_thresold;           /* "sof_thresold" (sic), default 0.6 */
        int ldpc_trials;              /* "ldpc_trials", default 10 */
        int ldpc_batch;               /* dvbs2::simd_type::SIZE of the build being replaced: frames per BBFrameLDPC::decode call sharing one early exit
                                         (16 with SSE4.1, 1 generic); frames wait in the handle until a group is full, as in process_s2 */
        const int8_t *lut_bits;       /* HOST: constellation_t::make_lut(lut_resolution)'s soft bits [x][y][bit] (module_dvbs2_demod.cpp:123-124) */
        const float *lut_phase_error; /* HOST: the same table's phase errors [x][y] (:114-115); both are copied at create */
        int lut_resolution;           /* 256 */
    } sdhip_dvbs2_cfg;
    typedef struct sdhip_dvbs2_stats
    {
        uint64_t samples_in;   /* baseband samples consumed */
        uint64_t plframes;     /* frames the PL synchroniser emitted */
        uint64_t bbframes;     /* BBFRAMEs written */
        float snr, peak_snr;   /* "snr", "peak_snr": M2M4 estimate over the last frame's slots */
        float freq_hz;         /* "freq": rad_to_hz(current_freq / final_sps, final_samplerate) -- the rotator's share, as in the module */
        float pll_freq;        /* S2PLLBlock::getFreq() behind the last frame, rad / symbol */
        float ldpc_trials;     /* "ldpc_trials" of the last decoder group (max trials when it did not converge) */
        float bch_corrections; /* "bch_corrections" of the last frame */
        int detected_modcod, detected_shortframes, detected_pilots; /* S2BBToSoft's PLS decode of the last frame (-1 before the first) */
        uint32_t pll_lanes, pll_rerun, pll_forced, pll_serial_frames, pll_branch_tries; /* the frame PLL's schedule, summed over the calls: lanes, lanes re-run from the
                                                                         exact predecessor state, boundaries let through unlocked, frames walked serially, whole-batch
                                                                         launches spent on the estimates' frequency branch */
    } sdhip_dvbs2_stats;
    void sdhip_dvbs2_cfg_default(sdhip_dvbs2_cfg *cfg);
    void *sdhip_dvbs2_demod_create(const sdhip_dvbs2_cfg *cfg); /* NULL on error; the messages are the module's / get_dvbs2_cfg's */
    void sdhip_dvbs2_demod_destroy(void *h);
    int sdhip_dvbs2_demod_bbframe_bytes(void *h);
    /* Host-buffer path (what the pipeline module calls): append samples; whole batches are processed as they fill, flush processes the rest. */
    int sdhip_dvbs2_demod_push(void *h, const void *iq, size_t nsamples, int fmt);
    int sdhip_dvbs2_demod_flush(void *h);
    /* Pop up to cap_frames BBFRAMEs (bbframe_bytes each). Returns the frame count. */
    int64_t sdhip_dvbs2_demod_pull(void *h, uint8_t *bbframes, size_t cap_frames);
    /* Device-resident path: nsamples baseband samples already in HBM -> the BBFRAMEs this call completes, written to d_bbframes (device).
       Returns the frame count, <0 on error. Stream state carries across calls. */
    int64_t sdhip_dvbs2_demod_process_dev(void *h, const void *d_iq, size_t nsamples, int fmt, uint8_t *d_bbframes, size_t cap_frames);
    /* The same behind the clock recovery: nsyms clock-recovered symbols (complex floats, device) enter at the rotator / PL synchroniser. */
    int64_t sdhip_dvbs2_demod_symbols_dev(void *h, const float *d_syms, size_t nsyms, uint8_t *d_bbframes, size_t cap_frames);
    int sdhip_dvbs2_demod_get_stats(void *h, sdhip_dvbs2_stats *st);

    /* ---- stream-parallel sharding of one recording over several devices / ranks (shard.hip) ------------------------------------------
       Host logic only. The reference runs ONE stream (a thread per module, src-core/pipeline/pipeline_run.cpp:72-104); cutting a recording in time is what
       N devices add, and these entry points are what makes the N chunks end in the single stream's CADU list: plan the sample ranges, find where a
       chunk's soft stream continues its predecessor's (so that its decoder starts on the single stream's Viterbi block grid), drop the frames two
       neighbours both decoded. bench.py --gpus N (torch.distributed, one process per GPU) and the plugin's "hip_devices" key use them. */
    typedef struct sdhip_shard_range
    {
        uint64_t read_start; /* first sample the chunk reads (own_start - overlap, clipped at 0) */
        uint64_t own_start;  /* first sample of the chunk's own range */
        uint64_t stop;       /* one past its last sample */
    } sdhip_shard_range;
    /* samples a chunk reads in front of its own range: the lock-in times of the stages from the modules' loop constants (shard.hip) */
    uint64_t sdhip_shard_overlap(const sdhip_demod_cfg *demod, const sdhip_fec_cfg *fec);
    /* its parts: out3 = {samples until a cold-started demodulator is locked, soft bytes a cold-started decoder needs in front of the first frame that counts,
       the decoder's block in soft bytes} */
    int sdhip_shard_lockin(const sdhip_demod_cfg *demod, const sdhip_fec_cfg *fec, uint64_t *out3);
    /* n_samples cut into `world` contiguous ranges whose boundaries are multiples of `align` samples */
    int sdhip_shard_plan(uint64_t n_samples, int world, uint64_t overlap, int align, sdhip_shard_range *out);
    /* prev_tail: the predecessor's LAST n_prev soft bytes; head: this chunk's first n_head soft bytes (both int8 as psk_demod writes them, q = 1 byte per
       symbol for BPSK, 2 otherwise). Both chunks demodulated the overlap's samples: finds *lag_symbols = the index in `head` of the symbol that follows the
       predecessor's last one, and *turn = the quarter turns (0..3; 0 / 2 for BPSK) this chunk's constellation is rotated by against the predecessor's,
       searching lags within `radius` symbols of `expect` (radius <= 0: everywhere). *agreement = fraction of equal hard decisions at the best lag (the two
       demodulators saw the same noise: ~1 at the right lag, ~0.5 elsewhere). Returns 0, 1 if no lag reaches 0.9, <0 on error. */
    int sdhip_shard_align(const int8_t *prev_tail, size_t n_prev, const int8_t *head, size_t n_head, int q, int64_t expect, int64_t radius, int64_t *lag_symbols, int *turn,
                          float *agreement);
    /* drops_out[r] = frames to drop at the head of chunk r's frame list, from boundary frames alone: heads[r] / tails[r] = its first n_heads[r] / last
       n_tails[r] frames (at most `edge` each), counts[r] = how many it decoded. whole_frames != 0: frames are compared whole (decoders on the common block
       grid); 0: behind the 4-byte sync marker. An overlap of more than `edge` frames is an error, not a silent duplicate. */
    int sdhip_shard_stitch(const uint8_t *const *heads, const size_t *n_heads, const uint8_t *const *tails, const size_t *n_tails, const uint64_t *counts, int world, int frame_bytes,
                           size_t edge, int whole_frames, uint64_t *drops_out);

    /* ---- the first step behind the CADUs: CCSDS AOS virtual channels and M_PDU packet extraction (aos_demux.hip) -----------------
       What every instrument decoder of the reference does first with a .cadu stream: ccsds::ccsds_aos::parseVCDU on each frame
       (src-core/common/ccsds/ccsds_aos/vcdu.cpp:10-18), keep the frames of its virtual channel, feed them to ccsds::ccsds_aos::Demuxer::work
       (demuxer.cpp:67-201) -- here with the CADUs and the packets' payload bytes staying in HBM. */
    typedef struct sdhip_vcdu
    {
        uint8_t version;
        uint16_t spacecraft_id;
        uint8_t vcid;
        uint32_t vcdu_counter;
        uint8_t replay_flag;
    } sdhip_vcdu; /* ccsds_aos::VCDU, vcdu.h:11-18 */
    int sdhip_aos_parse_vcdu_dev(int device, const uint8_t *d_cadus, int cadu_bytes, int nframes, sdhip_vcdu *d_out);
    /* the frames whose VCID is `vcid`, in order, copied to d_out (device); d_index_out (device, may be NULL) = their indices in the input. Returns the count. */
    int64_t sdhip_aos_select_vcid_dev(int device, const uint8_t *d_cadus, int cadu_bytes, int nframes, int vcid, uint8_t *d_out, size_t cap_frames, int *d_index_out);
    typedef struct sdhip_aos_packet
    {
        uint8_t header[6]; /* CCSDSHeader::raw and its fields (src-core/common/ccsds/ccsds.cpp:12-22) */
        uint8_t version, type, secondary_header_flag, sequence_flag;
        uint16_t apid, packet_sequence_count, packet_length;
        uint32_t frame;          /* index (within the call) of the frame whose Demuxer::work call handed the packet out */
        uint32_t payload_size;   /* CCSDSPacket::payload.size(): what was gathered, which a damaged stream can leave short of packet_length + 1 */
        uint64_t payload_offset; /* of its bytes in the call's payload pool */
    } sdhip_aos_packet;
    /* Demuxer(mpdu_data_size, hasInsertZone, insertZoneSize, secondaryHeaderExtendsPkt), demuxer.h:35: one handle per virtual channel, its state (a packet
       spanning frames, a header split across frames) carried across calls */
    void *sdhip_aos_demux_create(int device, int mpdu_data_size, int has_insert_zone, int insert_zone_size, int secondary_header_extends_pkt);
    void sdhip_aos_demux_destroy(void *h);
    /* nframes CADUs of ONE virtual channel (device), in order -> the packets Demuxer::work hands out for them, in order: packets_out (HOST table), their payload
       bytes gathered into d_payload (DEVICE pool, packet k at payload_offset). Returns the packet count; *payload_bytes_out = pool bytes used.    After a capacity error (packet table / payload pool too small) the handle must be destroyed and created anew: the state machine has consumed the call's frames. */
    int64_t sdhip_aos_demux_work_dev(void *h, const uint8_t *d_cadus, int cadu_bytes, int nframes, sdhip_aos_packet *packets_out, size_t cap_packets, uint8_t *d_payload,
                                     size_t cap_payload, uint64_t *payload_bytes_out);

    /* ---- "meteor_lrpt_decoder" (the Viterbi27-based plugin decoders; lrpt_decoder.hip) ------------------------------------------
       METEORLRPTDecoderModule::process(), the classic branch (plugins/meteor_support/meteor/module_meteor_lrpt_decoder.cpp:201-262): soft symbols ->
       frame correlator on the encoded sync word (src-core/common/codings/correlator.cpp:68-176) -> rotate_soft -> viterbi::Viterbi27 -> NRZ-M if
       "diff_decode" -> derand_ccsds -> RS(255,223) x 4 (conventional basis) -> the CADUs whose four codewords decoded. "m2x_mode" (Viterbi1_2 +
       deframer, optionally deinterleaved) is not this entry's: such runs stay on the CPU module. */
    typedef struct sdhip_lrpt_cfg
    {
        int diff_decode; /* "diff_decode" (mandatory key of the module) */
        int device;
    } sdhip_lrpt_cfg;
    typedef struct sdhip_lrpt_stats
    {
        uint64_t soft_in;     /* soft bytes consumed */
        uint64_t frames_seen; /* frames the correlator placed */
        uint64_t frames_out;  /* CADUs written */
        float viterbi_ber;    /* "viterbi_ber": Viterbi27::ber() of the last frame */
        int correlator_lock;  /* "correlator_lock": the last frame sat at offset 0 */
        int cor;              /* the last frame's correlation (of 64) */
        int rs_errors[4];     /* the last frame's per-codeword error counts (-1 = uncorrectable); "rs_avg" = their mean */
    } sdhip_lrpt_stats;
    void sdhip_lrpt_cfg_default(sdhip_lrpt_cfg *cfg);
    void *sdhip_lrpt_create(const sdhip_lrpt_cfg *cfg);
    void sdhip_lrpt_destroy(void *h);
    /* Host-buffer path: append n soft bytes (.soft wire format); every complete frame is decoded, an incomplete one stays pending. */
    int sdhip_lrpt_push(void *h, const int8_t *soft, size_t n);
    /* Pop up to cap_frames CADUs (1024 bytes each). Returns the count. */
    int64_t sdhip_lrpt_pull(void *h, uint8_t *cadu, size_t cap_frames);
    /* Device-resident path: n more soft bytes in HBM -> CADUs in d_cadu (device). Returns the frames written, < 0 on error. */
    int64_t sdhip_lrpt_process_dev(void *h, const int8_t *d_soft, size_t n, uint8_t *d_cadu, size_t cap_frames);
    int sdhip_lrpt_get_stats(void *h, sdhip_lrpt_stats *st);

    /* ---- misc ------------------------------------------------------------------------ */
    const char *sdhip_last_error(void);
    const char *sdhip_version(void);
    int sdhip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* SDHIP_H */
