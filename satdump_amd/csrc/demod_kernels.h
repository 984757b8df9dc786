// demod_kernels.h -- psk_demod DSP chain on gfx950: launch-side declarations.
//
// Feed-forward stages (format convert, rational resampler, RRC FIR, quantiser) are plain data-parallel
// kernels. The three feedback loops of the reference (AGC agc.cpp:25-39, Costas costas_loop.cpp:23-65,
// M&M clock_recovery_mm.cpp:52-121) are nonlinear per-sample recurrences; they run CHUNK-SPECULATIVE:
// one lane per chunk of the stream, every chunk but the first starting `warmup` samples early from a
// default state, and a boundary certificate (state reached by the warm-up == state the previous chunk
// ended in) decides whether the chunk's output stands or the chunk is re-run from the exact state.
#pragma once
#include "common.h"

namespace sdhip
{
    struct cf32
    {
        float re, im;
    };

    constexpr int DEMOD_HIST = 128; // samples of history kept in front of every stage buffer (>= the longest resampler arm: 34 / rate taps, rate > 0.5)

    // chunk geometry of one speculative stage: chunk 0 = [0, L+W), chunk k>=1 = [W + k*L, W + (k+1)*L) (clipped to n)
    struct ChunkGeom
    {
        long long n;
        int L, W, K;
    };
    inline ChunkGeom make_geom(long long n, int L, int W)
    {
        ChunkGeom g{n, L, W, 1};
        if (n > (long long)L + W)
            g.K = 1 + (int)((n - (L + W) + L - 1) / L);
        return g;
    }
    __host__ __device__ inline long long chunk_begin(const ChunkGeom &g, int k) { return k == 0 ? 0 : (long long)g.W + (long long)k * g.L; }
    __host__ __device__ inline long long chunk_end(const ChunkGeom &g, int k)
    {
        const long long e = (long long)g.W + (long long)(k + 1) * g.L;
        return e < g.n ? e : g.n;
    }

    // ---- format conversion (baseband_interface.h:172-199) + iq_swap --------------------------------
    void launch_convert(const void *in, int fmt, int iq_swap, long long n, cf32 *out, hipStream_t st);

    // ---- frequency shift (dsp::FreqShiftBlock, freq_shift.cpp:18-46 -> VOLK's rotator2: out = in * phase, phase *= phase_delta per sample,
    // the phase renormalised every 512 samples of a call and at the end of every call; the reference calls it once per source buffer) ----
    struct RotState
    {
        float re, im; // the rotator's phase
    };
    // exact: one sequential lane, the generic kernel's float operations in its order, `calls` of buf_len samples (pos0 = samples of the
    // current call already consumed by earlier launches)
    void launch_rotator_seq(const cf32 *x, cf32 *y, long long n, RotState *state, float dre, float dim, int buf_len, int pos0, hipStream_t st);
    // chunk-parallel mode: closed form, one thread per sample. Phase = (absolute sample index) * f in 64-bit fixed-point turns (exact,
    // no drift), magnitude = the sawtooth |phase_delta|^(position inside the 512-sample run) the reference's renormalisation leaves
    // (up to 3e-5: it must be there for the 1e-5 contract); what the reference's float recurrence adds on top is a slowly varying phase
    // offset, which the carrier loop behind it tracks out.
    void launch_rotator_par(const cf32 *x, cf32 *y, long long n, long long abs0, unsigned long long f_fix, float mag_eps, int buf_len, hipStream_t st);

    // ---- Doppler correction (dsp::DopplerCorrectBlock::work, src-core/common/dsp/utils/doppler_correct.cpp:41-63): out = in * (cosf(-phase), sinf(-phase));
    // phase += freq, wrapped into +-2 pi; freq = freq * (1 - alpha) + target * alpha, the target set anew after every source buffer from the pass
    // prediction (:68-93: SGP4 on the host -- the caller supplies it, sdhip_demod_doppler_targets) ----
    struct DopState
    {
        float phase, freq;
    };
    // exact: one sequential lane, every float / double operation where the reference has it. targets[k] = the target during the k-th source buffer that
    // STARTS inside this launch (the buffer in progress at the launch's first sample runs on target_cur); pos0 = samples of that buffer already consumed
    void launch_doppler_seq(const cf32 *x, cf32 *y, long long n, DopState *state, float alpha, float target_cur, const float *targets, int buf_len, int pos0, hipStream_t st);
    // chunk-parallel mode: the recurrence in closed form, in double: within a source buffer freq_i = target + (f0 - target) beta^i, phase_i = ph0 + i target +
    // (f0 - target)(1 - beta^i) / alpha. starts[b] = {ph0, f0, target} of the b-th buffer touched by this launch (host, double recurrence from buffer to buffer).
    // What it leaves out is the float rounding of the reference's phase accumulation (a random walk of ~1e-4 rad per buffer): the carrier loop behind tracks it.
    struct DopStart
    {
        double ph0, f0, target;
    };
    void launch_doppler_par(const cf32 *x, cf32 *y, long long n, const DopStart *starts, double alpha, int buf_len, int pos0, hipStream_t st);

    // ---- DC block (correct_iq.cpp:27-31): acc = acc * (1 - alpha) + x * alpha; y = x - acc, alpha = 1e-4 ------------
    struct DcState
    {
        float acc_re, acc_im;
    };
    // one sequential lane: exact mode and the unit op (bit for bit the reference)
    void launch_dcblock_seq(const cf32 *x, cf32 *y, long long n, DcState *state, hipStream_t st);
    // chunk-parallel: the recurrence is linear, so the accumulator at every chunk start follows from an affine scan evaluated
    // in double (launch_dc_partial: per chunk B_k = sum beta^(len-1-i) alpha x_i; the host chains acc_{k+1} = beta^len acc_k +
    // B_k over the K chunks); a lane per chunk then runs the reference's float recurrence from that start value
    // (launch_dcblock). The float trajectory wanders ~2e-6 |acc| (rms) around the exact-arithmetic one (two roundings of
    // 3e-8 |acc| per step, remembered for 1/alpha steps), so a boundary is certified within 1e-5 |acc|, not bit for bit.
    struct DcParams
    {
        const DcState *starts; // accumulator at the start of every chunk
    };
    void launch_dc_partial(const cf32 *x, const ChunkGeom &g, double *partial /* 2 doubles per chunk */, hipStream_t st);
    void launch_dcblock(const cf32 *x, cf32 *y, const ChunkGeom &g, const DcParams &p, const DcState *start0, DcState *spec, DcState *endst, const int *redo,
                        int nredo, hipStream_t st);

    // ---- rational resampler (rational_resampler.cpp:43-64), fully parallel ---------------------------
    // out[m] for m in [0, nout): global output index m0+m; input index/phase follow inc=(m*decim)/interp, ctr=(m*decim)%interp
    struct ResampParams
    {
        int interp, decim, ntaps;
        const float *bank; // [interp][ntaps] device
    };
    // x points at input sample 0 of this call (history at negative indices); phase0 = d_ctr, inc0 = carried inc
    // hist = the DEMOD_HIST samples preceding x[0]: x - DEMOD_HIST when the history sits in front of a stage buffer, a separate
    // buffer when x is the caller's own cf32 input read in place; nothing at or past x[nin] is read
    void launch_resample(const cf32 *x, const cf32 *hist, long long nin, const ResampParams &p, int ctr0, int inc0, cf32 *y, long long nout, hipStream_t st);

    // ---- AGC -----------------------------------------------------------------------------------------
    struct AgcParams
    {
        float rate, reference, max_gain, init_gain;
        int fast; // 1: chunk-parallel mode's arithmetic (sd_sqrt_fast, fma) in the stand-alone stage too
        // chunk start gains from the affine scan (launch_agc_partial + the engine's chain over the chunks); nullptr = every lane warms up from init_gain
        const float *starts = nullptr;
        int input_mag = 0; // 1: ndsp::AGCFastBlock (dsp/agc/agc_fast.cpp:37-55): gain += rate * (reference - sqrtf(re^2 + im^2 of the INPUT) * gain)
    };
    // The AGC recurrence (agc.cpp:25-39 / dsp/agc/agc.cpp:22-39) is, for a positive gain, g <- min((1 - rate |x|) g + rate reference, max_gain): a clamped affine map of
    // the gain whose coefficients depend on the INPUT only. Such maps compose -- (a2, b2, c2) o (a1, b1, c1) = (a2 a1, a2 b1 + b2, min(a2 c1 + b2, c2)) for a2 >= 0 --
    // so the gain at every chunk start follows from one pass over the samples (this kernel: the composed map of every chunk, in double) and a chain over the K chunks
    // (host). What the float recurrence of the reference adds is rounding noise of ~3e-8 sqrt(1 / (2 rate |x|)) relative around that exact-arithmetic trajectory: the
    // boundary certificate (k_agc_verdict) compares the scan's start value with the predecessor lane's float end state.
    // partial: 4 doubles per chunk = {a, b, c, valid} (valid = 0: some sample had rate |x| > 1, the map is not monotone there -- the engine falls back to warm-ups)
    void launch_agc_partial(const cf32 *x, const ChunkGeom &g, const AgcParams &p, double *partial, hipStream_t st);
    struct AgcState
    {
        float gain;
    };
    // redo == nullptr: speculative pass over all chunks (spec/endst written). redo != nullptr: re-run chunks redo[0..nredo)
    // from endst[k-1].
    // optional checkpoint rows of a chunk stage (experimental early exit of re-run lanes, see k_chunks): ck = K * per_chunk states
    struct ChunkCkpt
    {
        void *ck = nullptr;
        int per_chunk = 0, len = 0; // checkpoints per chunk, samples between them (multiple of 8)
        float tol_a = 0, tol_b = 0; // Stage::close() windows
        unsigned long long *work = nullptr; // optional device counters {re-run lanes, pieces run, pieces of the full chunks}
    };
    void launch_agc(const cf32 *x, cf32 *y, const ChunkGeom &g, const AgcParams &p, const AgcState *start0, AgcState *spec, AgcState *endst,
                    const int *redo, int nredo, hipStream_t st, const ChunkCkpt &ck = ChunkCkpt());

    // ---- AGC + 31-tap RRC FIR as one lane-per-chunk stage (see AgcFirStage): the lane filters the AGC samples it produces ----------
    constexpr int AGCFIR_NT = 31;
    struct AgcFirParams
    {
        AgcParams agc;
        float taps[AGCFIR_NT + 1]; // reversed taps as launch_fir takes them (FIRBlock order, fir.cpp:30); kernel argument: scalar registers
    };
    struct AgcFirState
    {
        float gain;
        float lag[4];                 // gain at the end of each of the last four 8-sample blocks (lag[3]: 32 samples ago)
        float w[2 * (AGCFIR_NT - 1)]; // the last NT - 1 AGC outputs (re, im), oldest first
    };
    void launch_agc_fir(const cf32 *x, cf32 *y, const ChunkGeom &g, const AgcFirParams &p, const AgcFirState *start0, AgcFirState *spec, AgcFirState *endst,
                        const int *redo, int nredo, hipStream_t st);

    // ---- decimating FIR stage of the power-of-two pre-decimator (decimating_fir.cpp:47-89 inside power_decim.cpp:58-77) ----------
    // y[m] = sum_j x[inc0 + m*decim - (ntaps-1) + j] * rtaps[j], j ascending (oldest sample first, mul and add rounded separately);
    // samples at negative indices come from hist[ntaps + index] (the last ntaps samples of the previous call)
    void launch_decim_fir(const cf32 *x, const cf32 *hist, long long nin, const float *rtaps_dev, int ntaps, int decim, int inc0, cf32 *y, long long nout,
                          hipStream_t st);

    // ---- RRC FIR (fir.cpp:74-83), fully parallel; taps reversed on the host, ntaps <= 361 ----------------
    void launch_fir(const cf32 *x, cf32 *y, long long n, const float *rtaps_dev, int ntaps, hipStream_t st);

    // ---- Costas loop ---------------------------------------------------------------------------------
    struct CostasParams
    {
        float alpha, beta, fmin, fmax;
        int order;
        float init_freq; // warm-up start frequency
        int est_len;     // samples of the feed-forward start-phase estimate of a warm-up (0 = start at phase 0)
        int clip_branched; // 1: the ndsp CostasBlock's dsp::branched_clip (dsp/pll/costas.cpp:42) instead of the legacy block's branchless_clip
    };
    struct CostasState
    {
        float phase, freq;
    };
    void launch_costas(const cf32 *x, cf32 *y, const ChunkGeom &g, const CostasParams &p, const CostasState *start0, CostasState *spec, CostasState *endst,
                       const int *redo, int nredo, hipStream_t st, const ChunkCkpt &ck = ChunkCkpt());

    // ---- ndsp::CostasFastBlock (dsp/pll/costas_fast.cpp:15-89, the registry's "costas_fast_cc", dsp_flowgraph_register.cpp:294): the VCO as a complex
    // number turned by small-angle updates (no sine / cosine in the loop), phase and rate phasors renormalised by the bit-trick inverse square root every 65th
    // sample. ONE sequential lane (the reference's float operations in its order: bit-exact), state resident on the device across calls.
    struct CostasFastParams
    {
        float alpha, beta, fmin, fmax;
        float lim_min_re, lim_min_im, lim_max_re, lim_max_im; // freq_limit_min_cpx / freq_limit_max_cpx (costas_fast.h:44-45)
        int order;
        // chunk-parallel schedule only (a warm-up lane's start): the loop frequency to start from, the samples of the feed-forward start-phase estimate,
        // renorm_ctr at sample 0 of this call (the counter follows the stream's sample count: a lane starting at sample i carries (ctr_base + i) % 65)
        float init_freq;
        int est_len;
        unsigned ctr_base;
    };
    struct CostasFastState
    {
        float freq, pha_re, pha_im, fre_re, fre_im;
        unsigned ctr; // renorm_ctr
        // not the block's: the least distance of `freq` from its limits at the renormalisations this lane has run (negative: the limiter acted). `freq` is a pure
        // accumulator -- it feeds nothing but the limiter's comparison --, so lanes that merge in (pha, fre) keep whatever offset in it they started with: the
        // engine's bit-exact hand-off needs every limiter decision of a lane to be safe against that offset (DemodEngine::costas_fast_stage)
        float margin;
    };
    void launch_costas_fast(const cf32 *x, cf32 *y, long long n, const CostasFastParams &p, CostasFastState *state_dev, hipStream_t st);
    // the same loop lane-per-chunk (k_chunks: warm-up from init_freq / the start-phase estimate, spec[k] = state at the chunk start, endst[k] behind it; redo as for launch_costas)
    void launch_costas_fast_chunks(const cf32 *x, cf32 *y, const ChunkGeom &g, const CostasFastParams &p, const CostasFastState *start0, CostasFastState *spec, CostasFastState *endst,
                                   const int *redo, int nredo, hipStream_t st);

    // ---- AGC + RRC filter + Costas loop as ONE lane-per-chunk stage (see k_afc) -----------------------------------------------------
    // The lane that produces the filtered samples of a chunk also runs the carrier loop over them: the filter output never goes to
    // memory (16 B per sample of HBM traffic less), and the loop's dependent chain (sincos in double, ~30 operations deep per sample)
    // runs interleaved with the filter's independent multiply-adds of the following blocks instead of alone on its SIMD.
    // Chunk geometry = the Costas stage's (g.W = the Costas warm-up): the clock recovery behind it keeps indexing rot[] by it. A lane
    // starts w_agc + g.W samples in front of its chunk: AGC alone over the first w_agc samples (its window fills on the way), then AGC
    // + filter + a feed-forward start-phase estimate over est_len samples, then all three stages without stores up to the chunk start.
    struct AfcParams
    {
        AgcFirParams af;
        CostasParams cos;
        int w_agc; // samples of AGC-only warm-up in front of the Costas warm-up (multiple of 64)
    };
    struct AfcState
    {
        AgcFirState af;
        CostasState cos;
    };
    struct AfcCkpt // what a re-run lane compares itself with (AgcFirStage's and CostasStage's own early-exit rules)
    {
        float gain, lag3, phase, freq;
    };
    struct AfcCkptCfg
    {
        AfcCkpt *ck = nullptr;
        int per_chunk = 0, len = 0;
        float tol_phase = 0, tol_freq = 0;
    };
    // fast: the chunk-parallel mode's arithmetic (float sine / cosine, hardware square root, fused multiply-adds: see sd_sincosf_fast); false =
    // every float operation rounded where the reference's is (exact mode)
    void launch_afc(const cf32 *x, cf32 *y, const ChunkGeom &g, const AfcParams &p, const AfcState *start0, AfcState *spec, AfcState *endst, const int *redo,
                    int nredo, hipStream_t st, const AfcCkptCfg &ck, bool fast);

    // ---- carrier-tracking PLL (has_carrier, pll_carrier_tracking.cpp:23-66): same state layout and chunk scheme as the Costas loop,
    // one stable point per turn (no frame ambiguity)
    struct PllParams
    {
        float alpha, beta, fmin, fmax;
        float init_freq;       // warm-up start frequency
        const float *atan_tab; // device: 257 arctangents of i / 255 (design::atan_table)
    };
    void launch_pll(const cf32 *x, cf32 *y, const ChunkGeom &g, const PllParams &p, const CostasState *start0, CostasState *spec, CostasState *endst,
                    const int *redo, int nredo, hipStream_t st, const ChunkCkpt &ck = ChunkCkpt());

    // ---- M&M clock recovery + quantiser ------------------------------------------------------------------
    struct MmParams
    {
        float omega_gain, mu_gain, omega_mid, omega_limit, init_mu;
        const float *bank; // [128][8] device
        int oqpsk;         // DelayOneImag folded into the read (delay_one_imag.cpp:20-27)
        // rotation of the Costas output per Costas chunk (quarter turns for order 4, half turns for order 2,
        // eighth turns for order 8), indexed by the Costas stage's chunk geometry
        ChunkGeom cg;
        const int *rot;
        int order;
        int cap; // output capacity per chunk (symbols)
        float tol_omega; // hand-off window of the loop's RATE state, samples per symbol (the early exit of a re-run lane tests it like the boundary certificate does)
        int fast_syms;   // warm-up gear shift: symbols run with mu_gain * fast_mult and the omega term frozen
        float fast_mult;
        int q8, q8_bpsk; // q8: the symbols are stored as the module's int8 soft symbols (2 bytes per symbol row entry) instead of floats
        int fast;        // chunk-parallel mode's arithmetic: fused multiply-adds in the interpolator (exact mode: 0)
        // loop: 0 = Mueller & Muller (clock_recovery_mm.cpp), 1 = Gardner (clock_recovery_gardner.cpp: a second interpolation half a symbol back, the
        // zero-crossing sample; MmState::p_0T holds its last symbol). Same lanes, same hand-off certificate (time of the next symbol, rate).
        // clip_float: Gardner's two clips as dsp::branched_clip on floats (the ndsp block, dsp/clock_recovery/clock_recovery_gardner.cpp:115,131) instead of
        // the legacy block's BRANCHLESS_CLIP in double (common/dsp/clock_recovery/clock_recovery_gardner.cpp:84,96).
        // back: samples a lane's window reaches behind inc - 7 (Gardner: floor(omega_max / 2) + 1, at most MM_BACK_MAX; M&M: 0)
        // loop 2 = ndsp::MMClockRecoveryFastBlock<complex_t> (dsp/clock_recovery/clock_recovery_mm_fast.cpp:66-163, "fast_clock_recovery_mm_cc"): the M&M detector on a
        // LINEAR interpolation between samples inc - 7 and inc - 6 of the window, the rate term updated every fifth symbol. One sequential lane only (the cadence
        // counter rides in MmState::upd_cnt).
        int loop, clip_float, back;
        // tap (tests only, sdhip_demod_set_tap): the symbol rows receive, instead of the symbol, the interpolation's position on the arm grid -- (inc * 128 + arm) as a
        // 64-bit integer in the symbol's eight bytes -- so a test can tell the symbols two trajectories computed on the same arm from the ones a neighbouring arm gave
        int tap;
    };
    constexpr int MM_BACK_MAX = 17;
    struct MmState
    {
        float mu, omega;
        cf32 p_2T, p_1T, p_0T, c_2T, c_1T, c_0T;
        long long inc; // position in this call's sample index space
        unsigned upd_cnt, pad; // MmParams::loop == 2 only: omega_upd_cnt (clock_recovery_mm_fast.h:44)
    };
    // what the boundary certificate compares: the timing state (the delay lines follow from it and the data)
    struct MmCert
    {
        float mu, omega;
        long long inc;
    };
    struct MmCkpt // checkpoint of a clock-recovery lane: its state when the block ending at a multiple of MM_CK_SAMPLES has been fed
    {
        float mu, omega;
        long long inc;
        int cnt, pad;
    };
    // counts: 2 ints per chunk = {symbols emitted inside the chunk, extra symbols computed past its end (0..2, stored right after)}
    // spec_c / end_c: compact copies of spec / endst for the host
    void launch_mm(const cf32 *x, cf32 *sym_scratch, int *counts, const ChunkGeom &g, const MmParams &p, const MmState *start0, MmState *spec, MmState *endst,
                   MmCert *spec_c, MmCert *end_c, const int *redo, int nredo, hipStream_t st, MmCkpt *ck = nullptr, int ck_per_chunk = 0,
                   float ck_tol = 0.0f); // ck: optional per-chunk checkpoint rows (experimental early exit of re-run lanes, see k_mm)
    constexpr int MM_CK_SAMPLES = 1024;   // input samples between the checkpoints of a clock-recovery lane
    // compaction + quantiser (module_psk_demod.cpp:199-213): seg = 2 ints per chunk {first, count}: chunk k's symbols
    // [first, first+count) of its scratch row go to offsets[k]
    void launch_quantize(const cf32 *sym_scratch, const int *seg, const long long *offsets, int K, int cap, int bpsk, int8_t *soft, long long soft_cap,
                         float *syms, long long syms_cap, hipStream_t st);
    // fast_clock_recovery_mm_cc lane per (chunk, cadence) (k_mmfast): rows [K][6][cap], spec / endst / counts [K][6] (slot 5: re-runs); redo lanes start from redo_start[i]
    // (chunk 0 -- W + L samples, one lane -- has its own row of cap0 symbols behind the others)
    void launch_mmfast(const cf32 *x, cf32 *rows, int *counts, const ChunkGeom &g, int cap, int cap0, const MmParams &p, const MmState *start0, MmState *spec, MmState *endst,
                       const int *redo, const MmState *redo_start, int nredo, hipStream_t st);
    void launch_mmfast_gather(const cf32 *rows, const int *sel, const long long *offs, const int *counts, int K, int cap, cf32 *out, hipStream_t st);
    // the same compaction for scratch rows written by launch_mm with MmParams::q8 (sym_scratch then holds 2-byte entries)
    void launch_compact8(const cf32 *sym_scratch, const int *seg, const long long *offsets, int K, int cap, int bpsk, int8_t *soft, long long soft_cap,
                         hipStream_t st);
    // rotated copy of the last `cnt` Costas outputs (history for the next call), incl. the per-chunk rotation
    void launch_derotate(cf32 *x, long long n, const ChunkGeom &cg, const int *rot, int order, hipStream_t st);
    void launch_tail_copy(const cf32 *x, long long n, int cnt, const ChunkGeom &cg, const int *rot, int order, cf32 *out, hipStream_t st);
} // namespace sdhip
