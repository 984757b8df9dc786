// demod_engine.hip -- host side of psk_demod on the GPU: parameter derivation exactly as the reference
// module does it (module_demod_base.cpp:12-89, module_psk_demod.cpp:12-136), stage sequencing, the
// boundary certificates of the chunk-speculative loop stages, stream-state carry, and the C ABI.
#include "demod_kernels.h"
#include "dsp_design.h"
namespace sdhip
{
#include "power_decim_tables.inc"
}
#include "../../include/sdhip.h"
#include "host_pipe.h"
#include <atomic>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

namespace sdhip
{
    // |x|^M frequency estimate for the very first warm-up frequency (parallel; the loops themselves are exact and the boundary
    // certificates decide what stands: this only seeds the speculation). Two autocorrelation lags of z = x^order in one pass:
    // lag 1 is unambiguous over +-pi/order rad/sample but, on the matched filter's output, noisy and biased by the pulse shape
    // (NPP QPSK, 2 samples/symbol, 2^18 samples: 3e-3 rad/sample of spread against a loop bandwidth of 2e-3); lag ~2 symbols is
    // 5-10x tighter (4e-4) but ambiguous, so the host takes its branch next to the lag-1 value. partial: 4 doubles per block.
    // classic = 1: the plain x^order weighting (kept for order 8, where the 8th power is so noisy at working SNRs that neither
    // the longer lag nor the clamp buys accuracy -- measured -- and the start value is refined from the lanes' own loops instead).
    __global__ __launch_bounds__(256) void k_freq_est(const cf32 *x, long long n, int order, int lag, int classic, double *partial)
    {
        __shared__ double sre[256], sim[256], lre[256], lim[256];
        const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const long long stride = (long long)gridDim.x * blockDim.x;
        double are = 0, aim = 0, bre = 0, bim = 0;
        for (long long i = i0; i + lag < n; i += stride)
        {
            double ar = x[i].re, ai = x[i].im, br = x[i + 1].re, bi = x[i + 1].im, cr = x[i + lag].re, ci = x[i + lag].im;
            // z = (x/|x|)^order * min(|x|, 2): the phase raised to the order, weighted by the (clamped) magnitude. Weighting by
            // |x|^order lets a few hundred over-sized samples (the AGC's transient after a level step) outvote a million others.
            const double ma = sqrt(ar * ar + ai * ai), mb = sqrt(br * br + bi * bi), mc = sqrt(cr * cr + ci * ci);
            const double ia = ma > 1e-30 ? 1.0 / ma : 0.0, ib = mb > 1e-30 ? 1.0 / mb : 0.0, ic = mc > 1e-30 ? 1.0 / mc : 0.0;
            if (!classic)
                ar *= ia, ai *= ia, br *= ib, bi *= ib, cr *= ic, ci *= ic;
            for (int m = 1; m < order; m <<= 1)
            { // square log2(order) times: z^order
                const double tr = ar * ar - ai * ai, ti = 2 * ar * ai;
                ar = tr;
                ai = ti;
                const double ur = br * br - bi * bi, ui = 2 * br * bi;
                br = ur;
                bi = ui;
                const double vr = cr * cr - ci * ci, vi = 2 * cr * ci;
                cr = vr;
                ci = vi;
            }
            const double wa = classic ? 1.0 : (ma < 2.0 ? ma : 2.0), wb = classic ? 1.0 : (mb < 2.0 ? mb : 2.0), wc = classic ? 1.0 : (mc < 2.0 ? mc : 2.0);
            are += (br * ar + bi * ai) * (wa * wb); // z[n+1] * conj(z[n])
            aim += (bi * ar - br * ai) * (wa * wb);
            bre += (cr * ar + ci * ai) * (wa * wc); // z[n+lag] * conj(z[n])
            bim += (ci * ar - cr * ai) * (wa * wc);
        }
        sre[threadIdx.x] = are;
        sim[threadIdx.x] = aim;
        lre[threadIdx.x] = bre;
        lim[threadIdx.x] = bim;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1)
        {
            if ((int)threadIdx.x < s)
            {
                sre[threadIdx.x] += sre[threadIdx.x + s];
                sim[threadIdx.x] += sim[threadIdx.x + s];
                lre[threadIdx.x] += lre[threadIdx.x + s];
                lim[threadIdx.x] += lim[threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
        {
            partial[4 * blockIdx.x] = sre[0];
            partial[4 * blockIdx.x + 1] = sim[0];
            partial[4 * blockIdx.x + 2] = lre[0];
            partial[4 * blockIdx.x + 3] = lim[0];
        }
    }
    __global__ __launch_bounds__(256) void k_mean_abs(const cf32 *x, long long n, double *partial)
    {
        __shared__ double acc[256];
        const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const long long stride = (long long)gridDim.x * blockDim.x;
        double a = 0;
        for (long long i = i0; i < n; i += stride)
            a += sqrt((double)x[i].re * x[i].re + (double)x[i].im * x[i].im);
        acc[threadIdx.x] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1)
        {
            if ((int)threadIdx.x < s)
                acc[threadIdx.x] += acc[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            partial[blockIdx.x] = acc[0];
    }

    // =============================================================================================
    // Boundary certificates on the device. Every boundary k is judged from (state chunk k's warm-up reached, state chunk
    // k-1 ended in) alone, so the verdicts are one thread per boundary; what the sequential host loop carried along
    // (Costas frame rotation, symbol offsets) is a prefix sum. The host reads back four counters per round.
    // =============================================================================================
    struct VerdictOut
    {
        int nfail, inexact, rotated, overflow;
        int forced, loose; // loose: accepted, but outside the stage's TIGHT window (drives the warm-up adaptation, not a re-run)
        int wide, pad;     // wide: failed OUTSIDE the stage's wide window -- a lane that is not locked, as opposed to a hand-off a little off (carrier stages: re-run either way)
        long long total;
    };
    template <class S>
    __global__ void k_spec_from_prev(const int *list, int n, S *spec, const S *endst)
    { // a re-run chunk starts from the exact end state of its predecessor: that is the state its certificate holds now
        const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (i < n)
            spec[list[i]] = endst[list[i] - 1];
    }
    // CKPT mode: a Costas re-run starts from the predecessor's end state expressed in the frame the chunk's earlier run locked on
    // (phase + d * rot_unit), so that the re-run can merge with that run's checkpoints; the verdict then books the boundary as a
    // frame change of d like any accepted rotated boundary.
    __global__ void k_costas_spec_aligned(const int *list, int n, CostasState *spec, const CostasState *endst, double rot_unit)
    {
        const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (i >= n)
            return;
        const int k = list[i];
        const CostasState a = spec[k], b = endst[k - 1];
        const long long d = llround(((double)a.phase - (double)b.phase) / rot_unit);
        CostasState ns = b;
        double ph = (double)b.phase + (double)d * rot_unit;
        const double twopi = 2 * 3.14159265358979323846;
        while (ph > twopi)
            ph -= twopi;
        while (ph < -twopi)
            ph += twopi;
        ns.phase = (float)ph;
        spec[k] = ns;
    }
    // force: the round limit is reached (unlocked signal: every trajectory is noise-driven, there is no sequential one to be
    // faithful to): the boundary is let through as it is and counted
    __device__ __forceinline__ void verdict_fail(VerdictOut *vo, int *fails, int k, int force)
    {
        if (force)
            atomicAdd(&vo->forced, 1);
        else
            fails[atomicAdd(&vo->nfail, 1)] = k;
    }

    __global__ void k_agc_verdict(int K, const AgcState *spec, const AgcState *endst, float tol, VerdictOut *vo, int *fails, int force)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k < 1 || k >= K)
            return;
        const float a = spec[k].gain, b = endst[k - 1].gain;
        if (__float_as_uint(a) == __float_as_uint(b))
            return;
        if (fabsf(a - b) <= tol * fabsf(b))
            atomicAdd(&vo->inexact, 1);
        else
            verdict_fail(vo, fails, k, force);
    }
    // AGC + FIR stage: the boundary gain and the gain 32 samples in front of it (AgcFirState::lag[3]) -- both bit-equal means the
    // filter window behind the warm-up holds the predecessor's samples bit for bit
    __global__ void k_agcfir_verdict(int K, const AgcFirState *spec, const AgcFirState *endst, VerdictOut *vo, int *fails, int force)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k < 1 || k >= K)
            return;
        const float a = spec[k].gain, b = endst[k - 1].gain, a3 = spec[k].lag[3], b3 = endst[k - 1].lag[3];
        if (__float_as_uint(a) == __float_as_uint(b) && __float_as_uint(a3) == __float_as_uint(b3))
            return;
        if (fabsf(a - b) <= 1e-6f * fabsf(b) && fabsf(a3 - b3) <= 1e-6f * fabsf(b3))
            atomicAdd(&vo->inexact, 1);
        else
            verdict_fail(vo, fails, k, force);
    }
    __global__ void k_dc_verdict(int K, const DcState *spec, const DcState *endst, float tol, VerdictOut *vo, int *fails, int force)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k < 1 || k >= K)
            return;
        const DcState a = spec[k], b = endst[k - 1];
        if (__float_as_uint(a.acc_re) == __float_as_uint(b.acc_re) && __float_as_uint(a.acc_im) == __float_as_uint(b.acc_im))
            return;
        const float m = fmaxf(fabsf(b.acc_re), fabsf(b.acc_im));
        if (fabsf(a.acc_re - b.acc_re) <= tol * m && fabsf(a.acc_im - b.acc_im) <= tol * m)
            atomicAdd(&vo->inexact, 1);
        else
            verdict_fail(vo, fails, k, force);
    }
    // dm[k] = quarter/half/eighth turns chunk k's frame is ahead of chunk k-1's (0 for a bit-exact or re-run boundary)
    // tol_tight: the hand-off window proper (the soft-symbol contract: a phase step of d rad is a relative symbol error of d) -- outside it the chunk is re-run from
    // the predecessor's exact state until it is back on the speculative trajectory (a checkpoint, typically the first: the step decays within ~tau ln(d / tol) samples);
    // tol_phase: the WIDE window -- outside it the lane had not locked at all (counted in VerdictOut::wide: what the stage's warm-up adaptation goes by)
    __global__ void k_costas_verdict(int K, const CostasState *spec, const CostasState *endst, double rot_unit, int rot_mod, double tol_phase, double tol_tight, double tol_freq,
                                     int *dm, VerdictOut *vo, int *fails, int force)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k >= K)
            return;
        int d_out = 0;
        if (k >= 1)
        {
            const CostasState a = spec[k], b = endst[k - 1];
            if (!(__float_as_uint(a.phase) == __float_as_uint(b.phase) && __float_as_uint(a.freq) == __float_as_uint(b.freq)))
            {
                // the loop's stable points are rot_unit apart: accept a lock on another one and rotate it back
                const double dphi = (double)a.phase - (double)b.phase;
                const long long d = llround(dphi / rot_unit);
                const double resid = dphi - (double)d * rot_unit;
                const bool in_wide = fabs(resid) < tol_phase && fabs((double)a.freq - (double)b.freq) < tol_freq;
                if (in_wide && (fabs(resid) < tol_tight || force))
                {
                    d_out = (int)(((d % rot_mod) + rot_mod) % rot_mod);
                    if (d_out != 0)
                        atomicAdd(&vo->rotated, 1);
                    atomicAdd(&vo->inexact, 1);
                }
                else
                {
                    if (!in_wide)
                        atomicAdd(&vo->wide, 1);
                    verdict_fail(vo, fails, k, force); // re-run continues in the previous chunk's frame: dm = 0
                }
            }
        }
        dm[k] = d_out;
    }
    // Fused AGC + filter + Costas stage: one verdict per boundary = the AGC + filter rule (gain and the gain 32 samples earlier, bit-equal
    // or within 1e-6) AND the Costas rule (bit-equal, or inside the windows on one of the loop's stable points: dm = the frame change).
    __global__ void k_afc_verdict(int K, const AfcState *spec, const AfcState *endst, double rot_unit, int rot_mod, double tol_phase, double tol_tight, double tol_freq, int *dm,
                                  VerdictOut *vo, int *fails, int force)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k >= K)
            return;
        int d_out = 0;
        if (k >= 1)
        {
            const float ga = spec[k].af.gain, gb = endst[k - 1].af.gain, la = spec[k].af.lag[3], lb = endst[k - 1].af.lag[3];
            const CostasState a = spec[k].cos, b = endst[k - 1].cos;
            const bool agc_same = __float_as_uint(ga) == __float_as_uint(gb) && __float_as_uint(la) == __float_as_uint(lb);
            const bool agc_ok = agc_same || (fabsf(ga - gb) <= 1e-6f * fabsf(gb) && fabsf(la - lb) <= 1e-6f * fabsf(lb));
            const bool cos_same = __float_as_uint(a.phase) == __float_as_uint(b.phase) && __float_as_uint(a.freq) == __float_as_uint(b.freq);
            bool ok = agc_ok, in_wide = agc_ok;
            if (ok && !cos_same)
            {
                const double dphi = (double)a.phase - (double)b.phase;
                const long long d = llround(dphi / rot_unit);
                const double resid = dphi - (double)d * rot_unit;
                in_wide = fabs(resid) < tol_phase && fabs((double)a.freq - (double)b.freq) < tol_freq;
                ok = in_wide && (fabs(resid) < tol_tight || force); // (a forced boundary inside the wide window still books its frame change)
                if (ok)
                {
                    d_out = (int)(((d % rot_mod) + rot_mod) % rot_mod);
                    if (d_out != 0)
                        atomicAdd(&vo->rotated, 1);
                }
            }
            if (!ok && !in_wide)
                atomicAdd(&vo->wide, 1);
            if (!ok)
                verdict_fail(vo, fails, k, force); // re-run continues in the previous chunk's frame: dm = 0
            else if (!(agc_same && cos_same))
                atomicAdd(&vo->inexact, 1);
        }
        dm[k] = d_out;
    }
    // start state of a re-run lane: the predecessor's exact end state; with checkpoints its carrier phase is expressed in the frame the
    // chunk's earlier run locked on (see k_costas_spec_aligned), so that the re-run can merge with that run's checkpoints
    __global__ void k_afc_spec_fix(const int *list, int n, AfcState *spec, const AfcState *endst, double rot_unit, int align)
    {
        const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (i >= n)
            return;
        const int k = list[i];
        AfcState ns = endst[k - 1];
        if (align)
        {
            const CostasState a = spec[k].cos, b = ns.cos;
            const long long d = llround(((double)a.phase - (double)b.phase) / rot_unit);
            double ph = (double)b.phase + (double)d * rot_unit;
            const double twopi = 2 * 3.14159265358979323846;
            while (ph > twopi)
                ph -= twopi;
            while (ph < -twopi)
                ph += twopi;
            ns.cos.phase = (float)ph;
        }
        spec[k] = ns;
    }
    __global__ void k_afc_gather_freq(int K, const AfcState *endst, float *out)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k < K)
            out[k] = endst[k].cos.freq;
    }
    // symbol hand-off at an M&M boundary (see DemodEngine::process): skip[k] symbols dropped at the head of chunk k, extra[k-1]
    // look-ahead symbols of chunk k-1 appended
    __global__ void k_mm_verdict(int K, const MmCert *spec, const MmCert *endst, const int *counts, double tol, double tol_tight, float tol_omega, int *skip, int *extra,
                                 VerdictOut *vo, int *fails, int force)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k >= K)
            return;
        if (k == 0)
        {
            skip[0] = 0;
            extra[K - 1] = 0;
            return;
        }
        int sk = 0, ex = 0;
        const MmCert a = spec[k], b = endst[k - 1];
        const bool same = __float_as_uint(a.mu) == __float_as_uint(b.mu) && __float_as_uint(a.omega) == __float_as_uint(b.omega) && a.inc == b.inc;
        if (!same)
        {
            bool ok = false;
            const double d = (double)(a.inc - b.inc) + ((double)a.mu - (double)b.mu);
            const double om = (double)b.omega;
            const long long r = llround(d / om);
            if (fabs(d - (double)r * om) < tol && fabsf(a.omega - b.omega) < tol_omega)
            {
                if (fabs(d - (double)r * om) >= tol_tight)
                    atomicAdd(&vo->loose, 1);
                if (r == 0)
                    ok = true;
                else if (r > 0 && r <= counts[2 * (k - 1) + 1])
                { // chunk k starts r symbols late: chunk k-1's look-ahead fills the gap
                    ex = (int)r;
                    ok = true;
                }
                else if (r < 0 && -r <= 2 && -r < counts[2 * k])
                { // chunk k starts r symbols early: its first symbols duplicate chunk k-1's last ones
                    sk = (int)-r;
                    ok = true;
                }
            }
            if (ok)
                atomicAdd(&vo->inexact, 1);
            else
                verdict_fail(vo, fails, k, force);
        }
        skip[k] = sk;
        extra[k - 1] = ex;
    }
    // Prefix sums over the K chunks, two launches of K/1024 blocks: tile sums, then every block adds the sums of the tiles in front
    // of it to its own scan. mode 0: rot[k] = (sum of dm[0..k]) mod rot_mod; mode 1: seg = {skip, count}, offs[k] = symbols in
    // front of chunk k, vo->total, vo->overflow.
    __device__ __forceinline__ long long chunk_scan_value(int mode, int k, const int *dm, const int *counts, const int *skip, const int *extra)
    {
        return mode == 0 ? (long long)dm[k] : (long long)(counts[2 * k] - skip[k] + extra[k]);
    }
    __device__ __forceinline__ long long block_scan_incl(long long v, long long *sh)
    { // inclusive scan over the 1024 threads of the block
        const int t = (int)threadIdx.x;
        sh[t] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1)
        {
            const long long o = t >= d ? sh[t - d] : 0;
            __syncthreads();
            sh[t] += o;
            __syncthreads();
        }
        return sh[t];
    }
    __global__ __launch_bounds__(1024) void k_chunk_scan_sums(int K, int mode, const int *dm, const int *counts, const int *skip, const int *extra, int cap,
                                                               long long *tile_sums, VerdictOut *vo)
    {
        __shared__ long long sh[1024];
        const int k = (int)(blockIdx.x * 1024 + threadIdx.x);
        long long v = 0;
        if (k < K)
        {
            v = chunk_scan_value(mode, k, dm, counts, skip, extra);
            if (mode == 1 && counts[2 * k] + 2 > cap)
                atomicExch(&vo->overflow, 1);
        }
        const long long incl = block_scan_incl(v, sh);
        if (threadIdx.x == 1023)
            tile_sums[blockIdx.x] = incl;
    }
    __global__ __launch_bounds__(1024) void k_chunk_scan_apply(int K, int mode, const int *dm, int rot_mod, int *rot, const int *counts, const int *skip,
                                                                const int *extra, const long long *tile_sums, int *seg, long long *offs, VerdictOut *vo)
    {
        __shared__ long long sh[1024];
        __shared__ long long base_sh;
        const int t = (int)threadIdx.x;
        // sum of the tiles in front of this one (a few hundred at most)
        long long part = 0;
        for (int b = t; b < (int)blockIdx.x; b += 1024)
            part += tile_sums[b];
        const long long all = block_scan_incl(part, sh);
        if (t == 1023)
            base_sh = all;
        __syncthreads();
        const long long base = base_sh;
        __syncthreads();
        const int k = (int)(blockIdx.x * 1024 + t);
        const long long v = k < K ? chunk_scan_value(mode, k, dm, counts, skip, extra) : 0;
        const long long incl = block_scan_incl(v, sh) + base;
        if (k < K)
        {
            if (mode == 0)
                rot[k] = (int)(incl % rot_mod);
            else
            {
                seg[2 * k] = skip[k];
                seg[2 * k + 1] = (int)v;
                offs[k] = incl - v;
                if (k == K - 1)
                    vo->total = incl;
            }
        }
    }

    // hist <- last nt samples of [hist | cur[0 .. ncur)] (one block; nt <= 1024)
    __global__ __launch_bounds__(1024) void k_hist_slide(cf32 *hist, int nt, const cf32 *cur, long long ncur)
    {
        const int i = (int)threadIdx.x;
        cf32 v{0, 0};
        if (i < nt)
        {
            const long long src = (long long)i + ncur - nt; // index into cur; negative: still inside the old history
            v = src >= 0 ? cur[src] : hist[nt + src];
        }
        __syncthreads();
        if (i < nt)
            hist[i] = v;
    }

    // the ndsp PSK demodulator (satdump::ndsp::PSKDemodHierBlock, dsp/hier/psk_demod.h) runs on the same engine: same loop kernels,
    // other stage order (RRC -> AGC -> M&M -> Costas at one sample per symbol), no resampler, no quantiser; what the legacy
    // module's configuration cannot say rides here
    struct NdspExt
    {
        bool on = false;
        double samplerate = 0, symbolrate = 0, rrc_gain = 1, rrc_alpha = 0.35; // doubles in RRC_Block (dsp/filter/rrc.h:17-21)
        float agc_reference = 1.0f, agc_gain = 1.0f, agc_max_gain = 65536.0f, rec_omega = 0.0f, pll_freq_limit = 1.0f;
        // the DVB-S2 demodulator's front (plugins/dvb_support/dvbs2/module_dvbs2_demod.cpp:98-105): BaseDemodModule's stages, the RRC filter and the
        // clock recovery like psk_demod's -- and NO Costas loop: carrier recovery happens per frame behind the PL synchroniser (sdhip_s2_pll_dev).
        // The legacy stage order, the symbols leave as floats.
        bool skip_costas = false;
        // ndsp: ONE member block of the chain on its own (the flowgraph's agc_cc / rrc_fir_cc / clock_recovery_mm_cc / costas_cc nodes):
        // 0 = the whole hier block, else SDHIP_NDSP_RRC_FIR / _AGC / _MM / _COSTAS
        int only = 0;
    };

    struct DemodEngine
    {
        sdhip_demod_cfg cfg;
        NdspExt nd;
        long long fir_drop = 0; // ndsp FIRBlock latency: the first ntaps outputs of the stream do not exist there (dsp/filter/fir.cpp:80-83)
        DevBuf<cf32> symtmp;    // ndsp: the clock recovery's symbols, input of the symbol-rate Costas loop
        hipStream_t stream = nullptr;
        // derived exactly like BaseDemodModule::initb / PSKDemodModule::init
        int d_buffer_size = 0;
        bool resample = false;
        float final_samplerate = 0, final_sps = 0;
        unsigned r_interp = 0, r_decim = 0;
        int r_ntaps = 0;
        bool is_bpsk = false, is_oqpsk = false;
        int order = 4, rot_mod = 4;
        double rot_unit = 0;
        int rrc_ntaps = 31;
        AgcParams agc_p{};
        CostasParams cos_p{};
        MmParams mm_p{};
        int tap_mode = 0; // tests only: sdhip_demod_set_tap
        double mm_windows_tight = 0.0, mm_windows_tol = 0.0; // the clock recovery's hand-off windows when a caller sets its own (demod_set_mm_windows); 0 = the defaults

        // power-of-two pre-decimator of SmartResampler (smart_resampler.cpp:15-29): chain of decimating FIRs, each with its
        // phase (`inc` of decimating_fir.cpp:60-86) and the last ntaps samples of its input carried across calls
        struct DecimStage
        {
            int decim = 1, ntaps = 0, inc = 0;
            DevBuf<float> d_taps;
            DevBuf<cf32> d_hist; // input[-ntaps .. -1] of the next call
        };
        std::vector<std::unique_ptr<DecimStage>> pd_stages;
        int pd_decim = 1;
        bool rational = true;

        // stream state
        bool started = false;
        int r_ctr = 0, r_inc = 0;
        AgcState agc_s{1.0f};
        CostasState cos_s{0.0f, 0.0f};
        MmState mm_s{};
        DcState dc_s{0, 0}, dc2_s{0, 0}; // dc_block in front / post_costas_dc behind the Costas loop
        // freq_shift (dsp::FreqShiftBlock): the rotator's phase increment as the reference rounds it, the phase (exact mode), the stream position
        float rot_dre = 1.0f, rot_dim = 0.0f;
        RotState rot_s{1.0f, 0.0f};
        long long rot_abs = 0;
        // Doppler correction (dsp::DopplerCorrectBlock): rotator state, the target in force, position inside the source buffer in progress (0 .. buffer
        // size: at the size, the next sample opens a new buffer), the targets handed in for the buffers to come
        DopState dop_s{0.0f, 0.0f};
        double dop_ph = 0.0, dop_f = 0.0; // the same state as the chunk-parallel mode carries it (double)
        float dop_target = 0.0f;
        int dop_pos = 0;
        std::deque<float> dop_queue;
        DevBuf<DopState> d_dop;
        DevBuf<float> d_dop_t;
        DevBuf<DopStart> d_dop_starts;
        unsigned long long rot_fix = 0;
        float rot_mag_eps = 0.0f;
        DevBuf<RotState> d_rot_state;
        // AGC and the RRC filter as one stage (AgcFirStage): the filter window travels with the lane state, on the device
        bool fuse_agc_fir = false;
        AgcFirParams af_p{};
        DevBuf<AgcFirState> d_af_spec, d_af_end, d_af_start;
        // ... and the Costas loop on the same lanes (k_afc): SDHIP_FUSE_COSTAS=0 keeps it a stage of its own (A/B switch)
        bool fuse_afc = false;
        DevBuf<AfcState> d_afc_spec, d_afc_end, d_afc_start;
        DevBuf<AfcCkpt> d_afc_ck;
        DevBuf<float> d_afc_freq, d_fe_tmp;
        // has_carrier: carrier-tracking PLL + its DC block between the RRC filter and the Costas loop
        PllParams cpll_p{};
        CostasState cpll_s{0.0f, 0.0f};
        DcState dcc_s{0, 0};
        long long w_cpll_learned = 0;
        DevBuf<float> d_atan;
        DevBuf<CostasState> d_cpll_spec, d_cpll_end, d_cpll_start, d_cpll_ck;
        std::vector<cf32> hist_in, hist_agc, hist_cos; // DEMOD_HIST samples each (history in front of stage inputs)

        // device
        DevBuf<cf32> bufA, bufB, symbuf, d_hist, d_hist_in;
        DevBuf<float> d_rrc, d_mmbank, d_rbank;
        DevBuf<AgcState> d_agc_spec, d_agc_end, d_agc_start;
        DevBuf<CostasState> d_cos_spec, d_cos_end, d_cos_start;
        CostasFastParams cf_p{};            // SDHIP_NDSP_COSTAS_FAST: the block's parameters, its state resident on the device
        DevBuf<CostasFastState> d_cf_state, d_cf_spec, d_cf_end;
        CostasFastState cf_s{0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 0u, 3.0e38f}; // host copy of the block's state behind the last call
        long long cf_total = 0;                                  // samples the block has seen (renorm_ctr follows them)
        DevBuf<int> d_cf_redo;
        // fast_clock_recovery_mm_cc lane per (chunk, cadence): mmfast_stage
        DevBuf<cf32> d_mf_rows;
        DevBuf<MmState> d_mf_spec, d_mf_end, d_mf_redo_start;
        DevBuf<int> d_mf_counts, d_mf_sel, d_mf_redo;
        DevBuf<long long> d_mf_offs;
        DevBuf<MmState> d_mm_spec, d_mm_end, d_mm_start;
        DevBuf<MmCert> d_mm_spec_c, d_mm_end_c; // what the host certificate reads (16 B per chunk instead of the 72-byte state)
        DevBuf<MmCkpt> d_mm_ck;                 // per-chunk checkpoints for the early exit of re-run lanes
        DevBuf<AgcState> d_agc_ck;              // ... of the AGC and Costas lanes (SDHIP_CKPT)
        DevBuf<CostasState> d_cos_ck;
        const bool use_ckpt = env_int("SDHIP_CKPT", 1) != 0; // early exit of re-run lanes (checkpoints); SDHIP_CKPT=0: a re-run lane runs its whole chunk
        long long w_mm_learned = 0, w_cos_learned = 0; // warm-up lengths this stream has been found to need (see the stages)
        long long w_mm_learned_own = 0;                // ... under a caller's own hand-off windows (demod_set_mm_windows)
        unsigned long long mm_calls = 0;               // clock-recovery stage calls of this stream so far
        DevBuf<unsigned long long> d_ck_work;   // {lanes, pieces run, pieces of full chunks} x {agc, costas}, SDHIP_DEBUG only
        void ck_report(const char *stage, int slot)
        {
            unsigned long long w[3];
            SD_HIP(hipMemcpy(w, d_ck_work.p + 3 * slot, sizeof(w), hipMemcpyDeviceToHost));
            if (w[0])
                fprintf(stderr, "[sdhip] %-6s early exit: %llu re-run lanes ran %llu of %llu pieces (%.0f %%)\n", stage, w[0], w[1], w[2], 100.0 * (double)w[1] / (double)w[2]);
        }
        DevBuf<DcState> d_dc, d_dc_spec, d_dc_end, d_dc_starts;
        DevBuf<double> d_dc_partial;
        DevBuf<int> d_redo, d_rot, d_dm, d_counts, d_seg, d_skip, d_extra;
        DevBuf<long long> d_offsets, d_tile_sums;
        DevBuf<double> d_partial;
        DevBuf<int8_t> d_soft_tmp;
        DevBuf<uint8_t> d_in_tmp;

        // host path (host_pipe.h): two pinned staging buffers filled by the copy pool, shipped and processed by a worker thread while the caller fills
        // the other one; results queue up for pull()
        static constexpr size_t HOST_BATCH = (size_t)64u << 20; // samples per shipped batch
        std::unique_ptr<HostPipe> pipe;
        PinBuf<int8_t> h_out;
        std::mutex out_mu;
        struct OutChunk
        {
            std::unique_ptr<int8_t[]> p;
            size_t n;
        };
        std::deque<OutChunk> out_queue; // one chunk per shipped batch (no reallocation of what is queued), oldest first
        size_t out_read = 0;           // bytes of the front chunk already pulled

        sdhip_demod_stats stats{};
        // what sdhip_demod_get_stats hands out while a call is in flight on another thread (the host path's worker): the snapshot the previous call left
        std::mutex stats_mu;
        sdhip_demod_stats stats_pub{};
        int stats_busy = 0; // calls in flight (under stats_mu; a count: a nested or second process() must not clear it early -- ADVICE r5)
        struct StatsScope
        {
            DemodEngine &e;
            explicit StatsScope(DemodEngine &en) : e(en)
            { // under the lock (a getter that has seen "not busy" is copying `stats` with the lock held), and the snapshot is taken BEFORE the call touches `stats`
                std::lock_guard<std::mutex> lk(e.stats_mu);
                if (e.stats_busy++ == 0)
                    e.stats_pub = e.stats;
            }
            ~StatsScope()
            {
                std::lock_guard<std::mutex> lk(e.stats_mu);
                if (--e.stats_busy == 0)
                    e.stats_pub = e.stats;
            }
        };

        static int fmt_bytes(int fmt) { return (fmt == SDHIP_FMT_CF32 || fmt == SDHIP_FMT_CS32) ? 8 : (fmt == SDHIP_FMT_CS16 ? 4 : 2); }

        explicit DemodEngine(const sdhip_demod_cfg &c, const NdspExt *ne = nullptr) : cfg(c)
        {
            if (ne)
                nd = *ne;
            SD_HIP(hipSetDevice(cfg.device));
            SD_HIP(hipStreamCreate(&stream));
            if (cfg.samplerate <= 0)
                throw HipError("Samplerate parameter must be present!");
            if (cfg.symbolrate <= 0)
                throw HipError("symbolrate must be present");
            // ---- BaseDemodModule ctor + initb (module_demod_base.cpp:22-25, 59-89)
            const long d_samplerate = (long)cfg.samplerate;
            const int d_symbolrate = (int)cfg.symbolrate;
            d_buffer_size = cfg.buffer_size > 0 ? cfg.buffer_size : std::min<int>(1000000, std::max<int>(8192 + 1, (int)(d_samplerate / 200)));
            float MIN_SPS = cfg.min_sps, MAX_SPS = cfg.max_sps;
            is_bpsk = cfg.constellation == SDHIP_BPSK;
            is_oqpsk = cfg.constellation == SDHIP_OQPSK;
            if (cfg.constellation != SDHIP_BPSK && cfg.constellation != SDHIP_QPSK && cfg.constellation != SDHIP_OQPSK && cfg.constellation != SDHIP_8PSK)
                throw HipError("Constellation type parameter must be present!");
            if (is_oqpsk)
            {
                MIN_SPS = 1.6f;
                MAX_SPS = 2.4f;
            }
            const float input_sps = (float)d_samplerate / (float)d_symbolrate;
            resample = input_sps > MAX_SPS || input_sps < MIN_SPS;
            const int range = (int)pow(10, (std::to_string(int(d_symbolrate)).size() - 1));
            final_samplerate = d_samplerate;
            if (cfg.custom_samplerate > 0) // "custom_samplerate", module_demod_base.cpp:73-74: the rate only -- the decision above stands
                final_samplerate = (long)cfg.custom_samplerate;
            else if (MAX_SPS == MIN_SPS)
                final_samplerate = d_symbolrate * MAX_SPS;
            else if (input_sps > MAX_SPS)
                final_samplerate = resample ? (round(d_symbolrate / range) * range) * MAX_SPS : d_samplerate;
            else if (input_sps < MIN_SPS)
                final_samplerate = resample ? d_symbolrate * MIN_SPS : d_samplerate;
            const float decimation_factor = d_samplerate / final_samplerate;
            if (resample)
                d_buffer_size *= ceil(decimation_factor);
            if (d_buffer_size > 8192 * 20)
                d_buffer_size = 8192 * 20;
            final_sps = final_samplerate / (float)d_symbolrate;
            if (nd.on)
            { // no resampler in the hier block; rec_blk "omega" = samplerate / symbolrate, a double narrowed to the block's float (psk_demod.h:224)
                if (resample)
                    throw HipError("ndsp chain: unexpected resample decision");
                final_samplerate = (float)nd.samplerate;
                final_sps = nd.rec_omega > 0 ? nd.rec_omega : (float)(nd.samplerate / nd.symbolrate);
            }
            if (input_sps < 1.0)
                throw HipError("Your sampling rate is too low!");

            if (resample)
            {
                // SmartResamplerBlock(input, final_samplerate, d_samplerate), smart_resampler.cpp:8-61
                unsigned interpolation = (unsigned)final_samplerate, decimation = (unsigned)d_samplerate;
                if (decimation > interpolation)
                {
                    const int best_power = (int)floor(log2(decimation / interpolation));
                    double rsamp_in = decimation, fout = interpolation, t;
                    if (best_power > 0)
                    { // PowerDecimatorBlock(best_decim): the plan's decimating FIR stages, power_decim.cpp:6-31
                        const int best_decim = std::min<int>(1 << best_power, 1 << 13);
                        rsamp_in = (double)decimation / (double)best_decim;
                        const PdPlan &plan = PD_PLANS[(int)log2(best_decim) - 1];
                        for (int i = 0; i < plan.nstages; i++)
                        {
                            const PdSet &set = PD_SETS[plan.stages[i].set];
                            if (set.count > 1024) // k_decim_fir stages its taps in 1024 floats of LDS, k_hist_slide runs 1024 threads
                                throw HipError("power-of-two decimator stage longer than 1024 taps");
                            auto st = std::make_unique<DecimStage>();
                            st->decim = plan.stages[i].decim;
                            st->ntaps = set.count;
                            std::vector<float> rev(set.count);
                            for (int j = 0; j < set.count; j++)
                            { // taps[(ntaps - 1) - j]: reversed like every FIR of the reference, decimating_fir.cpp:30
                                const unsigned bits = PD_TAP_BITS[set.offset + (set.count - 1 - j)];
                                memcpy(&rev[j], &bits, 4);
                            }
                            st->d_taps.reserve(rev.size());
                            SD_HIP(hipMemcpy(st->d_taps.p, rev.data(), rev.size() * sizeof(float), hipMemcpyHostToDevice));
                            st->d_hist.reserve((size_t)set.count + 1);
                            SD_HIP(hipMemset(st->d_hist.p, 0, ((size_t)set.count + 1) * sizeof(cf32)));
                            pd_stages.push_back(std::move(st));
                        }
                        pd_decim = best_decim;
                    }
                    rational = rsamp_in != fout;
                    if (rational)
                    {
                        while (modf(rsamp_in, &t) != 0 || modf(fout, &t) != 0)
                        {
                            rsamp_in *= 10;
                            fout *= 10;
                        }
                        interpolation = (unsigned)fout;
                        decimation = (unsigned)rsamp_in;
                    }
                }
                if (!rational)
                { // the ratio was an exact power of two: the pre-decimator is the whole resampler
                    r_interp = r_decim = 1;
                    r_ntaps = 0;
                }
                else
                {
                std::vector<float> bank;
                r_interp = interpolation;
                r_decim = decimation;
                r_ntaps = design::resampler_bank(r_interp, r_decim, bank);
                if (r_ntaps > DEMOD_HIST)
                    throw HipError("resampler filter longer than the history window");
                d_rbank.reserve(bank.size());
                SD_HIP(hipMemcpy(d_rbank.p, bank.data(), bank.size() * sizeof(float), hipMemcpyHostToDevice));
                }
            }
            // AGC (module_demod_base.cpp:207)
            agc_p.rate = cfg.agc_rate;
            agc_p.reference = nd.on ? nd.agc_reference : 1.0f;
            agc_p.max_gain = nd.on ? nd.agc_max_gain : 65536.0f;
            agc_p.init_gain = nd.on ? nd.agc_gain : 1.0f;
            agc_p.input_mag = nd.only == SDHIP_NDSP_AGC_FAST ? 1 : 0;
            agc_s.gain = agc_p.init_gain;
            // RRC (module_psk_demod.cpp:91)
            // ndsp: RRC_Block::set_cfg designs from its double members (dsp/filter/rrc.h:62-66)
            std::vector<float> rrc = nd.on ? design::rrc(nd.rrc_gain, nd.samplerate, nd.symbolrate, nd.rrc_alpha, cfg.rrc_taps)
                                           : design::rrc(1, final_samplerate, d_symbolrate, cfg.rrc_alpha, cfg.rrc_taps);
            rrc_ntaps = (int)rrc.size();
            if (rrc_ntaps > DEMOD_HIST || rrc_ntaps > 384)
                throw HipError("rrc_taps too large for the HIP path");
            std::vector<float> rrev(rrc.size());
            for (size_t j = 0; j < rrc.size(); j++)
                rrev[j] = rrc[rrc.size() - 1 - j]; // FIRBlock reverses its taps, fir.cpp:30
            d_rrc.reserve(rrev.size());
            SD_HIP(hipMemcpy(d_rrc.p, rrev.data(), rrev.size() * sizeof(float), hipMemcpyHostToDevice));
            // the 31-tap filter every pipeline of the path uses rides on the AGC lanes (SDHIP_FUSE_AGC_FIR=0: two kernels, A/B switch)
            fuse_agc_fir = rrc_ntaps == AGCFIR_NT && env_int("SDHIP_FUSE_AGC_FIR", 1) != 0 && !nd.on; // ndsp filters BEFORE the AGC
            fir_drop = nd.on ? rrc_ntaps : 0;
            if (fuse_agc_fir)
            {
                af_p.agc = agc_p;
                for (int j = 0; j < AGCFIR_NT; j++)
                    af_p.taps[j] = rrev[j];
                AgcFirState s0{};
                s0.gain = 1.0f;
                for (float &l : s0.lag)
                    l = 1.0f;
                d_af_start.reserve(1);
                SD_HIP(hipMemcpy(d_af_start.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
                fuse_afc = !cfg.has_carrier && env_int("SDHIP_FUSE_COSTAS", 1) != 0 && !nd.skip_costas;
                if (fuse_afc)
                {
                    AfcState a0{};
                    a0.af = s0;
                    a0.cos = CostasState{0.0f, 0.0f};
                    d_afc_start.reserve(1);
                    SD_HIP(hipMemcpy(d_afc_start.p, &a0, sizeof(a0), hipMemcpyHostToDevice));
                }
            }
            // carrier-tracking PLL (module_psk_demod.cpp:93-113, pll_carrier_tracking.cpp:8-21)
            if (cfg.has_carrier)
            {
                if (!is_bpsk)
                    throw HipError("For carrier mode, constellation must be BPSK!");
                if (!(cfg.carrier_pll_bw > 0))
                    throw HipError("Carrier PLL Bw parameter must be present!");
                design::costas_gains(cfg.carrier_pll_bw, cpll_p.alpha, cpll_p.beta); // same damping / denominator expression
                cpll_p.fmax = cfg.carrier_pll_max_offset;
                cpll_p.fmin = -cfg.carrier_pll_max_offset;
                const std::vector<float> at = design::atan_table();
                d_atan.reserve(at.size());
                SD_HIP(hipMemcpy(d_atan.p, at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice));
                cpll_p.atan_tab = d_atan.p;
                d_cpll_start.reserve(1);
            }
            // Costas (module_psk_demod.cpp:116-125)
            float costas_max_offset = cfg.has_carrier ? 0.2f : 1.0f; // "the offset in frequency should already be resolved on AM subcarriers"
            if (cfg.costas_max_offset_hz > 0)
                costas_max_offset = (float)(2.0 * design::PI * ((double)cfg.costas_max_offset_hz / (double)final_samplerate));
            order = is_bpsk ? 2 : (cfg.constellation == SDHIP_8PSK ? 8 : 4);
            rot_mod = order;
            rot_unit = 2.0 * design::PI / order;
            design::costas_gains(cfg.pll_bw, cos_p.alpha, cos_p.beta);
            if (nd.on)
                costas_max_offset = nd.pll_freq_limit; // CostasBlock "freq_limit", rad/sample (dsp/pll/costas.h:16, 33-34)
            cos_p.fmin = -costas_max_offset;
            cos_p.fmax = costas_max_offset;
            cos_p.clip_branched = nd.on ? 1 : 0;
            cos_p.order = order;
            cos_p.init_freq = 0.0f;
            if (nd.only == SDHIP_NDSP_COSTAS_FAST)
            { // CostasFastBlock::init (dsp/pll/costas_fast.h:39-61): the same gains, the limits' phasors from the host's cosf / sinf
                cf_p.alpha = cos_p.alpha;
                cf_p.beta = cos_p.beta;
                cf_p.fmin = -nd.pll_freq_limit;
                cf_p.fmax = nd.pll_freq_limit;
                cf_p.lim_min_re = cosf(cf_p.fmin);
                cf_p.lim_min_im = sinf(cf_p.fmin);
                cf_p.lim_max_re = cosf(cf_p.fmax);
                cf_p.lim_max_im = sinf(cf_p.fmax);
                cf_p.order = order;
                const CostasFastState s0{0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 0u, 3.0e38f};
                d_cf_state.reserve(1);
                SD_HIP(hipMemcpy(d_cf_state.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            }
            // one loop step moves the phase by at most fmax + beta + alpha: the kernel's phase wrap relies on that being under one turn
            if (!(costas_max_offset + cos_p.alpha + cos_p.beta < 6.0f))
                throw HipError("costas_max_offset / pll_bw too large for the HIP path (one loop step could exceed a full turn)");
            // M&M (module_psk_demod.cpp:134, clock_recovery_mm.cpp:10-27)
            std::vector<float> mmb;
            const int mmt = design::mm_bank(128, 8, mmb);
            if (mmt != 8)
                throw HipError("unexpected interpolator bank shape");
            d_mmbank.reserve(mmb.size());
            SD_HIP(hipMemcpy(d_mmbank.p, mmb.data(), mmb.size() * sizeof(float), hipMemcpyHostToDevice));
            mm_p.omega_gain = cfg.clock_gain_omega;
            mm_p.mu_gain = cfg.clock_gain_mu;
            mm_p.omega_mid = final_sps;
            mm_p.omega_limit = cfg.clock_omega_relative_limit * final_sps;
            mm_p.init_mu = cfg.clock_mu;
            mm_p.bank = d_mmbank.p;
            mm_p.oqpsk = is_oqpsk ? 1 : 0;
            mm_p.order = order;
            if (nd.only == SDHIP_NDSP_GARDNER)
            { // GardnerClockRecoveryBlock<complex_t> on the clock-recovery lanes (dsp/clock_recovery/clock_recovery_gardner.cpp:33-56)
                mm_p.loop = 1;
                mm_p.clip_float = 1;
                mm_p.back = (int)std::floor(((double)mm_p.omega_mid + std::fabs((double)mm_p.omega_limit)) / 2.0) + 1;
                if (mm_p.back > MM_BACK_MAX)
                    throw HipError("ndsp gardner: more than 32 samples per symbol are outside the window the HIP lanes carry");
            }
            if (nd.only == SDHIP_NDSP_MM_FAST)
                mm_p.loop = 2; // MMClockRecoveryFastBlock<complex_t> on one sequential lane (dsp/clock_recovery/clock_recovery_mm_fast.cpp:66-163)
            memset(&mm_s, 0, sizeof(mm_s));
            mm_s.mu = cfg.clock_mu;
            mm_s.omega = final_sps;
            mm_s.inc = 0;
            hist_in.assign(DEMOD_HIST, cf32{0, 0});
            hist_agc.assign(DEMOD_HIST, cf32{0, 0});
            hist_cos.assign(DEMOD_HIST, cf32{0, 0});
            d_hist.reserve(DEMOD_HIST);
            d_hist_in.reserve(DEMOD_HIST);
            d_agc_start.reserve(1);
            d_cos_start.reserve(1);
            d_mm_start.reserve(1);
            d_dc.reserve(1);
            d_partial.reserve(1024);
            if (cfg.freq_shift != 0)
            { // FreqShiftBlock::set_freq (freq_shift.cpp:37-43): phase_delta = (cos, sin)(hz_to_rad(shift, samplerate)) computed in double, stored as floats
                const double w = 2.0 * design::PI * ((double)(long)cfg.freq_shift / (double)d_samplerate);
                rot_dre = (float)std::cos(w);
                rot_dim = (float)std::sin(w);
                // closed form of the chunk-parallel mode: the increment's own angle in 2^-64 turns, its magnitude excess per step
                const double ang = std::atan2((double)rot_dim, (double)rot_dre) / (2.0 * design::PI);
                const long double turns = (long double)(ang < 0 ? ang + 1.0 : ang) * 18446744073709551616.0L;
                rot_fix = (unsigned long long)turns;
                rot_mag_eps = (float)(std::sqrt((double)rot_dre * rot_dre + (double)rot_dim * rot_dim) - 1.0);
                d_rot_state.reserve(1);
            }
            stats.final_sps = final_sps;
            stats.final_samplerate = final_samplerate;
            stats.buffer_size = d_buffer_size;
            stats.resample_interp = resample ? (int)r_interp : 0;
            stats.resample_decim = resample ? (int)r_decim : 0;
        }
        ~DemodEngine()
        {
            pipe.reset(); // the host path's threads end before the streams and the buffers they use go
            if (copy_stream)
                (void)hipStreamDestroy(copy_stream);
            if (stream)
                (void)hipStreamDestroy(stream);
        }

        void put_hist(cf32 *base, const std::vector<cf32> &h)
        {
            SD_HIP(hipMemcpyAsync(base - DEMOD_HIST, h.data(), DEMOD_HIST * sizeof(cf32), hipMemcpyHostToDevice, stream));
        }
        void get_hist(const cf32 *base, long long n, std::vector<cf32> &h)
        { // last DEMOD_HIST samples of [hist | data(n)]
            if (n >= DEMOD_HIST)
                SD_HIP(hipMemcpyAsync(h.data(), base + n - DEMOD_HIST, DEMOD_HIST * sizeof(cf32), hipMemcpyDeviceToHost, stream));
            else
            {
                std::vector<cf32> t(DEMOD_HIST);
                SD_HIP(hipMemcpyAsync(t.data(), base - (DEMOD_HIST - n), DEMOD_HIST * sizeof(cf32), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                h = t;
                return;
            }
            SD_HIP(hipStreamSynchronize(stream));
        }

        static long long env_int(const char *name, long long dflt)
        { // tuning overrides (experiments only; defaults are what ships)
            const char *v = getenv(name);
            return (v && *v) ? atoll(v) : dflt;
        }
        // Chunk length of one speculative stage: one lane per chunk, `lanes` lanes wanted. Measured (round 2's sweeps, profiles/history/r02/r02_sweeps.txt;
        // tools/ubench/lane_layout.hip): a lane streams its chunk in 64-byte blocks, and what the memory
        // system delivers for that pattern depends on how many lanes there are and how much each asks for at a time -- 65 k lanes
        // with 256 bytes per load group reach 4.5 TB/s (read + write) in a bare copy loop, 196 k lanes with 128 bytes 3.6. So every
        // stage runs ONE wave per SIMD (1024 SIMDs x 64 lanes, a little less so that no SIMD gets two) with 256-byte load groups
        // and the longest chunks that allows -- which also keeps the warm-up a small part of the chunk. (The Costas stage ran
        // three waves per SIMD with 128-byte groups while it was issue-bound on the two-call sincos: 14.0 -> 12.5 ms on MetOp.)
        // Chunks stay long enough (2048) that the warm-up overlap does not dominate.
        // SDHIP_LANES_AGC / _COSTAS / _MM override the targets (experiments only).
        enum StageKind { ST_AGC = 0, ST_COSTAS = 1, ST_MM = 2 };
        int pick_L(long long n, StageKind st) const
        {
            if (cfg.exact || (nd.only == SDHIP_NDSP_MM_FAST && st == ST_MM)) // (the _fast clock recovery's own lanes are mmfast_stage's; here it is one sequential lane)
                return 1 << 30;
            if (cfg.chunk_len <= 0 && getenv("SDHIP_CHUNK"))
                return (int)((env_int("SDHIP_CHUNK", 8192) + 7) / 8 * 8);
            if (cfg.chunk_len > 0)
                return (cfg.chunk_len + 7) / 8 * 8; // stage chunk boundaries stay multiples of 8 samples (64-byte blocks)
            static const char *lnames[3] = {"SDHIP_CHUNK_AGC", "SDHIP_CHUNK_COSTAS", "SDHIP_CHUNK_MM"};
            if (getenv(lnames[st]))
                return (int)((env_int(lnames[st], 8192) + 7) / 8 * 8);
            static const char *names[3] = {"SDHIP_LANES_AGC", "SDHIP_LANES_COSTAS", "SDHIP_LANES_MM"};
            // M&M: one wave per SIMD. Round 3 / 4 ran 98 304 lanes (one and a half waves: 13.0 ms against 15.5 at 65 280 on MetOp); since the symbol loop's
            // wave-uniform fast paths (round 5: 185 -> 86 instructions per symbol) a lone wave per SIMD is no slower per lane than two sharing one, and the
            // longer chunks of 65 280 lanes pay less warm-up: 11.2 ms against 13.3 (98 304) and 13.2 (130 560), parity 99.614 % against 99.592 % (profiles/history/r05/r05_d_ab_metop_ahrpt.txt)
            static const long long dflt[3] = {65280, 65280, 65280};
            static const long long min_len[3] = {2048, 2048, 2048};
            long long lanes = std::max<long long>(64, env_int(names[st], dflt[st]));
            long long L = (n + lanes - 1) / lanes;
            L = (L + 63) / 64 * 64;
            return (int)std::min<long long>(std::max<long long>(L, min_len[st]), 1 << 20);
        }

        // Certificate chain of one speculative stage. Chunk k stands iff its verdict kernel accepts (state its warm-up reached,
        // state chunk k-1 ended in). Every chunk that fails is re-run from the exact end state of its predecessor -- all failing
        // chunks of a round in ONE launch -- and the chain is judged again (a re-run changes that chunk's end state, which its
        // successor is then checked against), until nothing fails. Only the four counters of VerdictOut cross PCIe.
        DevBuf<VerdictOut> d_vout;
        PinBuf<VerdictOut> h_vout;
        DevBuf<int> d_fails;
        template <class Verdict, class SpecFix, class Launch>
        VerdictOut verify_fix(const char *stage, const int &K, Verdict verdict, SpecFix specfix, Launch relaunch)
        {
            return verify_fix(stage, K, verdict, specfix, relaunch, [](int) { return false; });
        }
        // respec(nfail): called when the FIRST judgement of the stage fails for more than an eighth of the chunks -- the
        // speculation's start values or warm-up length were off, not a few boundaries. It may re-launch the whole stage with better
        // ones (true = it did: judge again from scratch; it is then asked again if that judgement still fails as widely).
        template <class Verdict, class SpecFix, class Launch, class Respec>
        VerdictOut verify_fix(const char *stage, const int &K, Verdict verdict, SpecFix specfix, Launch relaunch, Respec respec, bool wide_counts = false)
        {
            int respecs = 0;
            d_vout.reserve(1);
            h_vout.reserve(1);
            d_fails.reserve((size_t)K + 1);
            unsigned reruns = 0, rounds = 0;
            // Every round makes at least the leftmost failing chunk of each run of failures exact, so a run of r consecutive
            // failing boundaries needs up to r rounds. Short runs (a glitch, a fade) are resolved exactly; where failures persist
            // beyond max_rounds the signal is not locked at all (noise before / after a pass) and the remaining boundaries are
            // let through (VerdictOut::forced): K rounds on pure noise would mean K launches of one lane each.
            // A short run that outlasts max_rounds -- the handful of chunks whose warm-ups straddle a level or frequency step of the signal
            // (the fused stage's warm-up spans several short chunks) -- is still resolved exactly, up to 4 x max_rounds: what marks the
            // unlocked case is that MANY boundaries keep failing.
            const unsigned max_rounds = (unsigned)env_int("SDHIP_MAX_ROUNDS", 4);
            int last_nf = 1 << 30;
            for (;;)
            {
                const int force = (rounds >= max_rounds && (last_nf > std::max(8, K / 64) || rounds >= 4 * max_rounds)) ? 1 : 0;
                SD_HIP(hipMemsetAsync(d_vout.p, 0, sizeof(VerdictOut), stream));
                verdict(d_vout.p, d_fails.p, force);
                SD_HIP(hipMemcpyAsync(h_vout.p, d_vout.p, sizeof(VerdictOut), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                const int nf = h_vout.p->nfail;
                // (carrier stages: a boundary inside the wide window but outside the tight one is re-run, not a sign of a short warm-up -- their verdicts count the
                // others in VerdictOut::wide; stages without the distinction leave it at zero and every failure counts)
                const int nwide = wide_counts ? h_vout.p->wide : nf;
                if (rounds == 0 && respecs < 8 && nwide + h_vout.p->loose > std::max(4, K / 8))
                {
                    respecs++;
                    if (respec(nwide + h_vout.p->loose))
                    {
                        if (getenv("SDHIP_DEBUG"))
                            fprintf(stderr, "[sdhip] %-6s %d of %d boundaries missed their (tight) window at first sight: stage re-launched\n", stage, nf + h_vout.p->loose, K);
                        continue;
                    }
                }
                if (nf == 0)
                    break;
                last_nf = nf;
                ++rounds;
                reruns += (unsigned)nf;
                specfix(d_fails.p, nf);
                relaunch(d_fails.p, nf);
            }
            stats.chunks_fixed += reruns;
            stats.chunks_inexact += (unsigned)h_vout.p->inexact;
            stats.chunks_rotated += (unsigned)h_vout.p->rotated;
            stats.chunks_forced += (unsigned)h_vout.p->forced;
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] %-6s chunks %d  re-run %u in %u round(s)  accepted-by-tolerance %d  let through unlocked %d\n", stage, K, reruns, rounds,
                        h_vout.p->inexact, h_vout.p->forced);
            return *h_vout.p;
        }

        // has_carrier front-end: the carrier-tracking PLL as a speculative chunk stage (one stable point per turn: no frame to
        // correct, the Costas verdict kernel with a 2 pi "rotation unit"), in -> out. The loop is linear while its detector
        // (arg(x) - phase, wrapped) stays off the wrap, so two trajectories on the same samples contract like exp(-1.414 bw t)
        // down to float noise; the windows are the Costas stage's.
        void carrier_pll_chunked(const cf32 *in, cf32 *out, long long n)
        {
            if (!started)
            { // start frequency of the warm-ups: lag-1 autocorrelation of the filtered signal (carrier and data both turn at it)
                const long long m = std::min<long long>(n, 1 << 20);
                ProfScope _ps("k_freq_est", stream);
                hipLaunchKernelGGL(k_freq_est, dim3(64), dim3(256), 0, stream, in, m, 1, 2, 0, d_partial.p);
                double part[256];
                SD_HIP(hipMemcpyAsync(part, d_partial.p, sizeof(part), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                double sr = 0, si = 0;
                for (int i = 0; i < 64; i++)
                    sr += part[4 * i], si += part[4 * i + 1];
                cpll_p.init_freq = std::min(std::max((float)std::atan2(si, sr), cpll_p.fmin), cpll_p.fmax);
            }
            else
                cpll_p.init_freq = cpll_s.freq;
            const long long w_cap = 1 << 20;
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)std::max(512.0, 24.0 / (1.414 * std::max(1e-5f, cfg.carrier_pll_bw)));
            W = std::max(W, w_cpll_learned);
            W = (std::min<long long>(W, w_cap) + 255) / 256 * 256;
            const int L = pick_L(n, ST_COSTAS);
            const double tol_phase = env_int("SDHIP_COSTAS_TOL_URAD", 10000) * 1e-6, tol_freq = env_int("SDHIP_COSTAS_TOL_NFREQ", 40000) * 1e-9;
            ChunkGeom g;
            ChunkCkpt ck;
            auto setup = [&](long long Wn) {
                g = make_geom(n, L, (int)Wn);
                d_cpll_spec.reserve(g.K);
                d_cpll_end.reserve(g.K);
                d_dm.reserve(g.K);
                if (use_ckpt)
                {
                    ck.len = 2048;
                    ck.per_chunk = L / ck.len + 1;
                    d_cpll_ck.reserve((size_t)g.K * ck.per_chunk);
                    ck.ck = d_cpll_ck.p;
                    ck.tol_a = (float)tol_phase;
                    ck.tol_b = (float)tol_freq;
                }
            };
            setup(W);
            SD_HIP(hipMemcpyAsync(d_cpll_start.p, &cpll_s, sizeof(cpll_s), hipMemcpyHostToDevice, stream));
            launch_pll(in, out, g, cpll_p, d_cpll_start.p, d_cpll_spec.p, d_cpll_end.p, nullptr, 0, stream, ck);
            verify_fix(
                "cpll", g.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_costas_verdict, dim3((g.K + 255) / 256), dim3(256), 0, stream, g.K, d_cpll_spec.p, d_cpll_end.p, 2.0 * design::PI, 1, tol_phase,
                                       tol_phase, tol_freq, d_dm.p, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    // a re-run starts from the predecessor's end state as it is: with one stable point per turn the earlier run's
                    // checkpoints are in the same frame (the early-exit test compares modulo 2 pi)
                    hipLaunchKernelGGL(k_spec_from_prev<CostasState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_cpll_spec.p, d_cpll_end.p);
                },
                [&](const int *redo, int nr) { launch_pll(in, out, g, cpll_p, d_cpll_start.p, d_cpll_spec.p, d_cpll_end.p, redo, nr, stream, ck); },
                [&](int) {
                    // many warm-ups missed: the start frequency was off (take the median of the lanes' end frequencies) or the
                    // warm-up is too short for this loop bandwidth (double it; the stream keeps the longer one)
                    std::vector<CostasState> es((size_t)g.K);
                    SD_HIP(hipMemcpyAsync(es.data(), d_cpll_end.p, es.size() * sizeof(CostasState), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    std::vector<float> fr(es.size());
                    for (size_t i = 0; i < es.size(); i++)
                        fr[i] = es[i].freq;
                    std::nth_element(fr.begin(), fr.begin() + fr.size() / 2, fr.end());
                    const float med = fr[fr.size() / 2];
                    if (std::fabs(med - cpll_p.init_freq) > 0.05f * cfg.carrier_pll_bw)
                        cpll_p.init_freq = med;
                    else if (cfg.warmup <= 0 && 2 * (long long)g.W <= w_cap)
                    {
                        w_cpll_learned = 2 * (long long)g.W;
                        setup(w_cpll_learned);
                    }
                    else
                        return false;
                    launch_pll(in, out, g, cpll_p, d_cpll_start.p, d_cpll_spec.p, d_cpll_end.p, nullptr, 0, stream, ck);
                    return true;
                });
            stats.chunks += g.K;
            SD_HIP(hipMemcpyAsync(&cpll_s, d_cpll_end.p + (g.K - 1), sizeof(cpll_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
        }

        // Chunk-parallel DC block (see demod_kernels.h): affine scan in double for the accumulator at every chunk start, then the
        // reference's float recurrence per chunk, certified against the previous chunk's end within 1e-5 |acc|. in -> out,
        // `carried` = the accumulator across calls.
        void dc_block_chunked(const cf32 *in, cf32 *out, long long n, DcState &carried)
        {
            const int L = pick_L(n, ST_AGC);
            const ChunkGeom g = make_geom(n, L, 0);
            d_dc_partial.reserve(2 * (size_t)g.K);
            d_dc_spec.reserve(g.K);
            d_dc_end.reserve(g.K);
            d_dc_starts.reserve(g.K);
            launch_dc_partial(in, g, d_dc_partial.p, stream);
            std::vector<double> part(2 * (size_t)g.K);
            SD_HIP(hipMemcpyAsync(part.data(), d_dc_partial.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            std::vector<DcState> starts((size_t)g.K);
            const double beta = (double)(1.0f - 0.0001f);
            double sr = carried.acc_re, si = carried.acc_im;
            for (int k = 0; k < g.K; k++)
            {
                starts[k] = DcState{(float)sr, (float)si};
                const double a = std::pow(beta, (double)(chunk_end(g, k) - chunk_begin(g, k)));
                sr = a * sr + part[2 * (size_t)k];
                si = a * si + part[2 * (size_t)k + 1];
            }
            starts[0] = carried; // chunk 0 starts from the carried state itself
            SD_HIP(hipMemcpyAsync(d_dc_starts.p, starts.data(), starts.size() * sizeof(DcState), hipMemcpyHostToDevice, stream));
            SD_HIP(hipMemcpyAsync(d_dc.p, &carried, sizeof(carried), hipMemcpyHostToDevice, stream));
            const DcParams dp{d_dc_starts.p};
            launch_dcblock(in, out, g, dp, d_dc.p, d_dc_spec.p, d_dc_end.p, nullptr, 0, stream);
            const float dc_tol = 1e-5f;
            verify_fix(
                "dc", g.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_dc_verdict, dim3((g.K + 255) / 256), dim3(256), 0, stream, g.K, d_dc_spec.p, d_dc_end.p, dc_tol, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    hipLaunchKernelGGL(k_spec_from_prev<DcState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_dc_spec.p, d_dc_end.p);
                },
                [&](const int *redo, int nr) { launch_dcblock(in, out, g, dp, d_dc.p, d_dc_spec.p, d_dc_end.p, redo, nr, stream); });
            stats.chunks += g.K;
            SD_HIP(hipMemcpyAsync(&carried, d_dc_end.p + (g.K - 1), sizeof(carried), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
        }

        // AGC + RRC filter + Costas loop as ONE speculative lane stage (k_afc), in -> out (the Costas output in per-chunk frames; cg =
        // its chunk geometry, d_rot the frames, as the clock recovery expects them from the stand-alone Costas stage).
        // Warm-up of a lane = the AGC's (24 gain / rate samples, AGC alone) followed by the Costas loop's (24 loop time constants, all
        // three stages); boundary certificate = both stages' rules (k_afc_verdict); re-runs start from the exact predecessor state and
        // stop at the first checkpoint at which they are back on the speculative run's trajectory.
        void afc_chunked(const cf32 *in, cf32 *out, long long n, ChunkGeom &cg)
        {
            // ---- AGC part of the warm-up (see the stand-alone AGC stage below for the reasoning)
            float g_est = agc_s.gain;
            if (!started)
            {
                const long long m = std::min<long long>(n, 1 << 16);
                ProfScope _ps("k_mean_abs", stream);
                hipLaunchKernelGGL(k_mean_abs, dim3(64), dim3(256), 0, stream, in, m, d_partial.p);
                double part[64];
                SD_HIP(hipMemcpyAsync(part, d_partial.p, sizeof(part), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                double sm = 0;
                for (double v : part)
                    sm += v;
                const double mean = sm / (double)m;
                if (mean > 1e-12)
                    g_est = (float)std::min((double)(agc_p.max_gain > 0 ? agc_p.max_gain : 65536.0f), (double)agc_p.reference / mean);
            }
            const double tau = std::max(1.0f, g_est) / std::max(1e-6f, cfg.agc_rate);
            long long Wa = cfg.warmup > 0 ? cfg.warmup : (long long)(24.0 * tau);
            Wa = env_int("SDHIP_W_AGC", Wa);
            Wa = std::min<long long>(std::max<long long>(Wa, 1024), 1 << 22);
            Wa = (Wa + 255) / 256 * 256;
            agc_p.init_gain = g_est;
            af_p.agc = agc_p;
            // ---- start frequency of the carrier loop's warm-ups: this stream's tracked frequency; on the very first call the M-th power
            // estimate (k_freq_est) over the first samples, which the stage's input must be filtered for: a plain FIR pass over that prefix
            if (!started)
            {
                const int classic = order > 4 ? 1 : 0;
                const long long m = std::min<long long>(n, classic ? 1 << 18 : 1 << 20);
                cos_p.init_freq = 0.0f;
                if (m > 4096)
                {
                    // the estimator wants what the stand-alone Costas stage would see: level-normalised, matched-filtered samples (on the raw
                    // samples a level step in the stream lets the strong part outvote the rest and the unambiguous lag-1 value goes wrong --
                    // found by tests/test_demod_emu_cpu.py's amplitude_step scenario). The AGC + filter stage over the prefix, its chunks
                    // speculative and unverified: this only seeds the warm-ups' start frequency.
                    d_fe_tmp.reserve(2 * ((size_t)m + 64) + 64);
                    cf32 *tmp = reinterpret_cast<cf32 *>(d_fe_tmp.p);
                    const ChunkGeom fg = make_geom(m, pick_L(m, ST_AGC), (int)Wa);
                    d_af_spec.reserve(fg.K);
                    d_af_end.reserve(fg.K);
                    launch_agc_fir(in, tmp, fg, af_p, d_af_start.p, d_af_spec.p, d_af_end.p, nullptr, 0, stream);
                    const int lag = (int)std::min(16.0, std::max(2.0, std::floor(2.0 * final_sps + 0.5)));
                    ProfScope _ps("k_freq_est", stream);
                    hipLaunchKernelGGL(k_freq_est, dim3(64), dim3(256), 0, stream, tmp, m, order, lag, classic, d_partial.p);
                    double part[256];
                    SD_HIP(hipMemcpyAsync(part, d_partial.p, sizeof(part), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    double sr = 0, si = 0, lr = 0, li = 0;
                    for (int i = 0; i < 64; i++)
                    {
                        sr += part[4 * i];
                        si += part[4 * i + 1];
                        lr += part[4 * i + 2];
                        li += part[4 * i + 3];
                    }
                    const double coarse = std::atan2(si, sr) / order;
                    double fine = coarse;
                    if (!classic && m > 4 * lag && (lr != 0 || li != 0))
                    {
                        const double step = 2.0 * design::PI / ((double)order * lag);
                        const double base = std::atan2(li, lr) / ((double)order * lag);
                        fine = base + step * std::floor((coarse - base) / step + 0.5);
                    }
                    if (getenv("SDHIP_DEBUG"))
                        fprintf(stderr, "[sdhip] costas start frequency: lag-1 %.6f, lag-%d %.6f rad/sample (%lld samples)\n", coarse, lag, fine, m);
                    cos_p.init_freq = std::min(std::max((float)fine, cos_p.fmin), cos_p.fmax);
                }
            }
            else
                cos_p.init_freq = cos_s.freq;
            const long long w_cos_cap = 1 << 20;
            // 20 loop time constants: the lane starts next to a stable point (feed-forward phase estimate over est_len samples) at the
            // stream's own frequency, so it reaches the float floor of two trajectories sooner than the 24 the stand-alone stage allows.
            // Measured: MetOp 17 GB on the GPU (profiles/history/r03/r03_a_ab_metop.txt) soft parity 0.99617 at 16 against 0.99612 at 24, 0.99591 at 12;
            // on the host twin with 8192-sample chunks (three times the boundaries) NPP's pll_bw 0.002 loses 0.27 % of the symbols at 16
            // and nothing at 20 (0.00393 against 0.00389 beyond 1e-5).
            const double taus = (double)env_int("SDHIP_COSTAS_TAUS", 20);
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)std::max(512.0, taus / (1.414 * std::max(1e-5f, cfg.pll_bw)));
            const long long w_first = (W + 255) / 256 * 256;
            W = std::max(W, w_cos_learned);
            W = env_int("SDHIP_W_COSTAS", W);
            W = (std::min<long long>(W, w_cos_cap) + 255) / 256 * 256;
            // two waves per SIMD: the stage is bound by its dependent chains (AGC sqrt, sincos in double), not by its loads -- measured
            // (MetOp, profiles/history/r03/r03_a_ab_metop.txt): 26.4 ms with 65 280 lanes, 21.3 with 98 304, 19.5 with 130 560 (226 VGPRs: two waves fit)
            // Round 6: one wave per SIMD instead where that is the faster plan. A lane's time is (warm-up + chunk) sequential samples at the per-sample pace of
            // its occupancy -- two waves sharing a SIMD each run ~1.3 times slower than one alone (MetOp 19.5 against 26.4 ms above = 1.31 with the chunk
            // lengths put in; GOES, profiles/r06_c_*: 3.35 against 3.17 ms) -- and the warm-up does not shrink with the chunk: on a 2 GiB stream (GOES: 5.6 k
            // samples of warm-up in front of 2 k-sample chunks at 130 560 lanes) half the lanes with chunks twice as long finish sooner, and hand off half as
            // often (GOES full size: float symbols within 1e-5 0.99155 -> 0.99231, symbols two or more arms off 5 812 -> 2 949 of 12.4 M).
            long long lanes_dflt = 130560;
            if (!cfg.exact && cfg.chunk_len <= 0)
            {
                const double w = (double)Wa + (double)W;
                const double cost2 = (w + std::max(2048.0, (double)n / 130560.0)) * 1.3, cost1 = w + std::max(2048.0, (double)n / 65280.0);
                if (cost1 < cost2)
                    lanes_dflt = 65280;
            }
            const long long lanes = std::max<long long>(64, env_int("SDHIP_LANES_AFC", lanes_dflt));
            int L = pick_L(n, ST_COSTAS);
            if (!cfg.exact && cfg.chunk_len <= 0 && !getenv("SDHIP_CHUNK") && !getenv("SDHIP_CHUNK_COSTAS") && !getenv("SDHIP_LANES_COSTAS"))
                L = (int)std::min<long long>(std::max<long long>(((n + lanes - 1) / lanes + 63) / 64 * 64, 2048), 1 << 20);
            L = (int)(((long long)L + 63) / 64 * 64); // chunk starts on whole load groups (the lane finds its chunk start on a group boundary)
            const double tol_phase = env_int("SDHIP_COSTAS_TOL_URAD", 10000) * 1e-6, tol_freq = env_int("SDHIP_COSTAS_TOL_NFREQ", 40000) * 1e-9;
            // the hand-off window proper (round 6; see costas_stage): a boundary between it and the wide window is re-run from the exact state and stops at the first
            // checkpoint at which it is back within it
            const double tol_tight = std::min(tol_phase, env_int("SDHIP_COSTAS_TIGHT_URAD", 10000) * 1e-6);
            AfcParams ap;
            AfcCkptCfg ck;
            // chunk-parallel mode: the stage's fast arithmetic (demod_kernels.hip, sd_sincosf_fast) -- inside the 1e-5 contract that mode is
            // held to; exact mode rounds every operation where the reference does. SDHIP_FAST_MATH=0: the exact arithmetic in both (A/B)
            const bool afc_fast = !cfg.exact && env_int("SDHIP_FAST_MATH", 1) != 0;
            auto setup = [&](long long Wn) {
                cos_p.est_len = (int)(std::min<long long>(env_int("SDHIP_COSTAS_EST", 256), Wn / 2) / 32 * 32);
                cg = make_geom(n, L, (int)Wn);
                ap.af = af_p;
                ap.cos = cos_p;
                ap.w_agc = (int)Wa;
                d_afc_spec.reserve(cg.K);
                d_afc_end.reserve(cg.K);
                ck.len = 2048;
                ck.tol_phase = (float)tol_tight;
                ck.tol_freq = (float)tol_freq;
                if (use_ckpt && !cfg.exact)
                {
                    ck.per_chunk = L / ck.len + 1;
                    d_afc_ck.reserve((size_t)cg.K * ck.per_chunk);
                    ck.ck = d_afc_ck.p;
                }
                d_rot.reserve(cg.K);
                d_dm.reserve(cg.K);
            };
            setup(W);
            // the carried start state lives on the device (filter window); its carrier part is the host's copy (kept in the stream's frame)
            SD_HIP(hipMemcpyAsync(&d_afc_start.p->cos, &cos_s, sizeof(cos_s), hipMemcpyHostToDevice, stream));
            launch_afc(in, out, cg, ap, d_afc_start.p, d_afc_spec.p, d_afc_end.p, nullptr, 0, stream, ck, afc_fast);
            verify_fix(
                "afc", cg.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_afc_verdict, dim3((cg.K + 255) / 256), dim3(256), 0, stream, cg.K, d_afc_spec.p, d_afc_end.p, rot_unit, rot_mod, tol_phase, tol_tight, tol_freq,
                                       d_dm.p, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    hipLaunchKernelGGL(k_afc_spec_fix, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_afc_spec.p, d_afc_end.p, rot_unit, use_ckpt ? 1 : 0);
                },
                [&](const int *redo, int nr) { launch_afc(in, out, cg, ap, d_afc_start.p, d_afc_spec.p, d_afc_end.p, redo, nr, stream, ck, afc_fast); },
                [&](int) {
                    // many warm-ups missed: start frequency off (take the median of the lanes' end frequencies) or warm-up too short for
                    // this signal's loop dynamics (double it; the stream keeps the longer one) -- see the stand-alone Costas stage
                    std::vector<float> fr((size_t)cg.K);
                    d_afc_freq.reserve(cg.K);
                    hipLaunchKernelGGL(k_afc_gather_freq, dim3((cg.K + 255) / 256), dim3(256), 0, stream, cg.K, d_afc_end.p, d_afc_freq.p);
                    SD_HIP(hipMemcpyAsync(fr.data(), d_afc_freq.p, fr.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    std::nth_element(fr.begin(), fr.begin() + fr.size() / 2, fr.end());
                    const float med = fr[fr.size() / 2];
                    if (std::fabs(med - cos_p.init_freq) > 0.05f * cfg.pll_bw)
                        cos_p.init_freq = med;
                    else if (cfg.warmup <= 0 && !getenv("SDHIP_W_COSTAS") && 2 * (long long)cg.W <= std::min(w_cos_cap, 4 * w_first))
                        w_cos_learned = 2 * (long long)cg.W; // (at most twice: a stream that still misses is not locked -- noise -- and every
                                                             // further doubling would only multiply the work of lanes that cannot merge)
                    else
                        return false;
                    setup(std::max<long long>(cg.W, w_cos_learned));
                    launch_afc(in, out, cg, ap, d_afc_start.p, d_afc_spec.p, d_afc_end.p, nullptr, 0, stream, ck, afc_fast);
                    return true;
                },
                true);
            stats.chunks += 2 * (unsigned)cg.K; // the chunks of two loop stages
            {
                const int nt = (cg.K + 1023) / 1024;
                d_tile_sums.reserve(nt);
                hipLaunchKernelGGL(k_chunk_scan_sums, dim3(nt), dim3(1024), 0, stream, cg.K, 0, d_dm.p, nullptr, nullptr, nullptr, 0, d_tile_sums.p, nullptr);
                hipLaunchKernelGGL(k_chunk_scan_apply, dim3(nt), dim3(1024), 0, stream, cg.K, 0, d_dm.p, rot_mod, d_rot.p, nullptr, nullptr, nullptr, d_tile_sums.p,
                                   nullptr, nullptr, nullptr);
            }
            int rot_last = 0;
            SD_HIP(hipMemcpyAsync(d_afc_start.p, d_afc_end.p + (cg.K - 1), sizeof(AfcState), hipMemcpyDeviceToDevice, stream));
            SD_HIP(hipMemcpyAsync(&agc_s, &d_afc_end.p[cg.K - 1].af.gain, sizeof(agc_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(&cos_s, &d_afc_end.p[cg.K - 1].cos, sizeof(cos_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(&rot_last, d_rot.p + (cg.K - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            // carry the loop state re-expressed in the stream's frame (rot 0), so that the next call starts unrotated
            if (rot_last != 0)
            {
                double ph = (double)cos_s.phase - rot_last * rot_unit;
                while (ph > 2 * design::PI)
                    ph -= 2 * design::PI;
                while (ph < -2 * design::PI)
                    ph += 2 * design::PI;
                cos_s.phase = (float)ph;
            }
            stats.freq_hz = (float)(((double)cos_s.freq / (2.0 * design::PI)) * (double)final_samplerate);
        }

        // ---- AGC with scanned start gains (see agc_stage): AIN -> OUT
        DevBuf<double> d_agc_partial;
        DevBuf<float> d_agc_starts;
        bool agc_scan_stage(const cf32 *AIN, cf32 *OUT, long long n, int L, double tau)
        {
            const ChunkGeom g = make_geom(n, L, 0);
            d_agc_partial.reserve(4 * (size_t)g.K);
            d_agc_starts.reserve(g.K);
            d_agc_spec.reserve(g.K);
            d_agc_end.reserve(g.K);
            launch_agc_partial(AIN, g, agc_p, d_agc_partial.p, stream);
            std::vector<double> part(4 * (size_t)g.K);
            SD_HIP(hipMemcpyAsync(part.data(), d_agc_partial.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            std::vector<float> starts((size_t)g.K);
            double gs = (double)agc_s.gain;
            bool valid = true;
            for (int k = 0; k < g.K; k++)
            {
                starts[k] = (float)gs;
                valid = valid && part[4 * (size_t)k + 3] != 0.0 && gs == gs;
                gs = std::fmin(part[4 * (size_t)k] * gs + part[4 * (size_t)k + 1], part[4 * (size_t)k + 2]);
            }
            starts[0] = agc_s.gain;
            if (!valid)
            { // rate |x| > 1 somewhere (the gain may have gone through zero: |gain| is not what the maps assume) or a NaN: this call runs on warm-ups
                if (getenv("SDHIP_DEBUG"))
                    fprintf(stderr, "[sdhip] agc    scan: a sample with rate * |x| > 1 (or a NaN) in this call: warm-up schedule instead\n");
                return false;
            }
            SD_HIP(hipMemcpyAsync(d_agc_starts.p, starts.data(), starts.size() * sizeof(float), hipMemcpyHostToDevice, stream));
            SD_HIP(hipMemcpyAsync(d_agc_start.p, &agc_s, sizeof(agc_s), hipMemcpyHostToDevice, stream));
            AgcParams ap = agc_p;
            ap.starts = d_agc_starts.p;
            ChunkCkpt agc_ck;
            const double steps = std::min((double)L, std::max(1.0, tau));
            const float tol = (float)std::min(1e-4, std::max(1e-6, 6.0 * 3e-8 * std::sqrt(steps)));
            if (use_ckpt)
            {
                agc_ck.len = 2048;
                agc_ck.per_chunk = L / agc_ck.len + 1;
                d_agc_ck.reserve((size_t)g.K * agc_ck.per_chunk);
                agc_ck.ck = d_agc_ck.p;
                agc_ck.tol_a = tol;
            }
            launch_agc(AIN, OUT, g, ap, d_agc_start.p, d_agc_spec.p, d_agc_end.p, nullptr, 0, stream, agc_ck);
            const int vb = (g.K + 255) / 256;
            verify_fix(
                "agc-scan", g.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_agc_verdict, dim3(vb), dim3(256), 0, stream, g.K, d_agc_spec.p, d_agc_end.p, tol, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    hipLaunchKernelGGL(k_spec_from_prev<AgcState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_agc_spec.p, d_agc_end.p);
                },
                [&](const int *redo, int nr) { launch_agc(AIN, OUT, g, ap, d_agc_start.p, d_agc_spec.p, d_agc_end.p, redo, nr, stream, agc_ck); });
            stats.chunks += g.K;
            SD_HIP(hipMemcpyAsync(&agc_s, d_agc_end.p + (g.K - 1), sizeof(agc_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            return true;
        }

        // Process n input samples resident on the device. Outputs go to d_soft / d_syms (device).
        // ---- AGC (speculative): AIN -> OUT (with the RRC filter on the same lanes when fuse_agc_fir)
        void agc_stage(const cf32 *AIN, cf32 *OUT, long long n)
        {
            // warm-up length ~ 24 time constants of the loop (tau = gain / rate samples), gain estimated from mean |x|
            float g_est = agc_s.gain;
            if (!started)
            {
                const long long m = std::min<long long>(n, 1 << 16);
                ProfScope _ps("k_mean_abs", stream);
                hipLaunchKernelGGL(k_mean_abs, dim3(64), dim3(256), 0, stream, AIN, m, d_partial.p);
                double part[64];
                SD_HIP(hipMemcpyAsync(part, d_partial.p, sizeof(part), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                double s = 0;
                for (double v : part)
                    s += v;
                const double mean = s / (double)m;
                if (mean > 1e-12)
                    g_est = (float)std::min((double)(agc_p.max_gain > 0 ? agc_p.max_gain : 65536.0f), (double)agc_p.reference / mean);
            }
            // tau = gain / rate samples; 24 tau of warm-up from the mean-based gain merge bit for bit with the previous chunk's
            // trajectory (13 tau would do within the 1e-6 tolerance; measured: the lane kernels are bound by their strided
            // HBM traffic, not by the chain -- a parallel-scan start value that cut W to 6 tau bought nothing net)
            const double tau = std::max(1.0f, g_est) / std::max(1e-6f, cfg.agc_rate);
            // (the ndsp chain: 14 -- inside the certificate's 1e-6 from a start value within tens of percent; at the block's rate 1e-4 the
            // warm-up IS the stage's time, (W + L) sequential steps per lane. SDHIP_AGC_TAUS overrides.)
            const double taus = (double)env_int("SDHIP_AGC_TAUS", nd.on ? 14 : 24);
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)(taus * tau);
            W = env_int("SDHIP_W_AGC", W);
            W = std::min<long long>(std::max<long long>(W, 1024), 1 << 22);
            W = (W + 255) / 256 * 256;
            agc_p.init_gain = g_est;
            agc_p.fast = (nd.on && !cfg.exact && env_int("SDHIP_FAST_MATH", 1) != 0) ? 1 : 0;
            int L = pick_L(n, ST_AGC);
            // ---- start gains by affine scan (round 6). A warm-up costs every lane W sequential steps; where that is a large part of the lane's work -- the ndsp
            // block's rate 1e-4 needs 290 k samples of it, 24.8 of the chain's 53.7 ms per 2^30 samples -- the gain at every chunk start is COMPUTED instead: the
            // recurrence is a clamped affine map of the gain whose coefficients depend on the input only (demod_kernels.h: launch_agc_partial), one pass composes
            // the map of every chunk in double, the host chains them over the K chunks, and the lanes run their chunks with no warm-up at all. The boundary
            // certificate then compares the scan's start value with the predecessor lane's float end state: they differ by the rounding noise the float
            // recurrence collects over min(L, tau) steps (~3e-8 sqrt(min(L, tau) / 2) relative), so its window is six of those instead of 1e-6; a boundary
            // outside it is re-run from the predecessor's state like any other; a call with a sample outside the scan's model (rate |x| > 1: the gain may pass through zero; a NaN) runs on warm-ups. The fused AGC + filter (+ Costas) stages of
            // the legacy chain keep their warm-up: at rate 1e-2 it is 15 % more samples of AGC-only work in a stage bound by the Costas loop's arithmetic, less
            // than a pass over the input for the scan would cost. SDHIP_AGC_SCAN=0/1 forces the choice.
            const bool scan_default = !fuse_agc_fir && cfg.warmup <= 0 && !getenv("SDHIP_W_AGC") && 4 * W > (long long)L;
            const bool use_scan = !cfg.exact && !fuse_agc_fir && env_int("SDHIP_AGC_SCAN", scan_default ? 1 : 0) != 0;
            if (use_scan && agc_scan_stage(AIN, OUT, n, L, tau))
                return; // (false: a sample outside the scan's model in this call -- the warm-up schedule below takes it)
            // a slow loop (the ndsp block's default rate 1e-4: 24 tau ~ 4e5 samples) on many short chunks would run K lanes over W + L
            // samples each -- a hundred times the stream through L2 / HBM for no gain in wall time, which is (W + L) sequential steps
            // either way: keep the chunk at least half the warm-up (work <= 3 n, still thousands of lanes on a bench-sized call)
            if (!cfg.exact && cfg.chunk_len <= 0 && !getenv("SDHIP_CHUNK") && !getenv("SDHIP_CHUNK_AGC"))
                L = (int)std::min<long long>(std::max<long long>(L, (W / 2 + 63) / 64 * 64), 1 << 22);
            const ChunkGeom g = make_geom(n, L, (int)W);
            stats.chunks += g.K;
            if (fuse_agc_fir)
            {
                // AGC + RRC filter in one pass: in -> OUT holds the FILTERED samples; the lane state (gain, last 30 AGC outputs)
                // stays on the device from call to call
                af_p.agc = agc_p;
                d_af_spec.reserve(g.K);
                d_af_end.reserve(g.K);
                launch_agc_fir(AIN, OUT, g, af_p, d_af_start.p, d_af_spec.p, d_af_end.p, nullptr, 0, stream);
                const int vb = (g.K + 255) / 256;
                verify_fix(
                    "agc+fir", g.K,
                    [&](VerdictOut *vo, int *fails, int force) {
                        hipLaunchKernelGGL(k_agcfir_verdict, dim3(vb), dim3(256), 0, stream, g.K, d_af_spec.p, d_af_end.p, vo, fails, force);
                    },
                    [&](const int *list, int nr) {
                        hipLaunchKernelGGL(k_spec_from_prev<AgcFirState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_af_spec.p, d_af_end.p);
                    },
                    [&](const int *redo, int nr) { launch_agc_fir(AIN, OUT, g, af_p, d_af_start.p, d_af_spec.p, d_af_end.p, redo, nr, stream); });
                SD_HIP(hipMemcpyAsync(d_af_start.p, d_af_end.p + (g.K - 1), sizeof(AgcFirState), hipMemcpyDeviceToDevice, stream));
                SD_HIP(hipMemcpyAsync(&agc_s, d_af_end.p + (g.K - 1), sizeof(agc_s), hipMemcpyDeviceToHost, stream)); // the gain is the state's first member
                SD_HIP(hipStreamSynchronize(stream));
            }
            else
            {
            d_agc_spec.reserve(g.K);
            d_agc_end.reserve(g.K);
            SD_HIP(hipMemcpyAsync(d_agc_start.p, &agc_s, sizeof(agc_s), hipMemcpyHostToDevice, stream));
            ChunkCkpt agc_ck;
            if (use_ckpt)
            {
                agc_ck.len = 2048;
                agc_ck.per_chunk = L / agc_ck.len + 1;
                d_agc_ck.reserve((size_t)g.K * agc_ck.per_chunk);
                agc_ck.ck = d_agc_ck.p;
                agc_ck.tol_a = 1e-6f;
                if (getenv("SDHIP_DEBUG"))
                {
                    d_ck_work.reserve(6);
                    SD_HIP(hipMemsetAsync(d_ck_work.p, 0, 6 * sizeof(unsigned long long), stream));
                    agc_ck.work = d_ck_work.p;
                }
            }
            launch_agc(AIN, OUT, g, agc_p, d_agc_start.p, d_agc_spec.p, d_agc_end.p, nullptr, 0, stream, agc_ck);
            const int vb = (g.K + 255) / 256;
            verify_fix(
                "agc", g.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_agc_verdict, dim3(vb), dim3(256), 0, stream, g.K, d_agc_spec.p, d_agc_end.p, 1e-6f, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    hipLaunchKernelGGL(k_spec_from_prev<AgcState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_agc_spec.p, d_agc_end.p);
                },
                [&](const int *redo, int nr) { launch_agc(AIN, OUT, g, agc_p, d_agc_start.p, d_agc_spec.p, d_agc_end.p, redo, nr, stream, agc_ck); });
            if (agc_ck.work)
                ck_report("agc", 0);
            SD_HIP(hipMemcpyAsync(&agc_s, d_agc_end.p + (g.K - 1), sizeof(agc_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            }
        }

        // ---- Costas (speculative, symmetry-corrected): A -> B; cg = the chunk geometry, d_rot the frame of every chunk
        // (sps: samples per symbol of the stage input -- the lag of the start-frequency estimate; rate_hz: its sample rate, for the stats)
        // carrier frequency a warm-up lane starts from, estimated over the head of a stream's first call (k_freq_est)
        float carrier_start_freq(const cf32 *A, long long n, double sps)
        {
            // carrier frequency for the warm-up start state: arg(sum z[n+L] conj(z[n])) / (order L), z = x^order, L ~ two
            // symbols, on the branch next to the (unambiguous, coarse) lag-1 value -- see k_freq_est
            const int classic = order > 4 ? 1 : 0;
            const long long m = std::min<long long>(n, classic ? 1 << 18 : 1 << 20);
            const int lag = (int)std::min(16.0, std::max(2.0, std::floor(2.0 * sps + 0.5)));
            ProfScope _ps("k_freq_est", stream);
            hipLaunchKernelGGL(k_freq_est, dim3(64), dim3(256), 0, stream, A, m, order, lag, classic, d_partial.p);
            double part[256];
            SD_HIP(hipMemcpyAsync(part, d_partial.p, sizeof(part), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            double sr = 0, si = 0, lr = 0, li = 0;
            for (int i = 0; i < 64; i++)
            {
                sr += part[4 * i];
                si += part[4 * i + 1];
                lr += part[4 * i + 2];
                li += part[4 * i + 3];
            }
            const double coarse = std::atan2(si, sr) / order;
            double fine = coarse;
            if (!classic && m > 4 * lag && (lr != 0 || li != 0))
            {
                const double step = 2.0 * design::PI / ((double)order * lag); // spacing of the lag-L branches
                const double base = std::atan2(li, lr) / ((double)order * lag);
                fine = base + step * std::floor((coarse - base) / step + 0.5);
            }
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] costas start frequency: lag-1 %.6f, lag-%d %.6f rad/sample (%lld samples)\n", coarse, lag, fine, m);
            float f = (float)fine;
            f = std::min(std::max(f, cos_p.fmin), cos_p.fmax);
            return f;
        }

        // ---- ndsp::MMClockRecoveryFastBlock<complex_t> (dsp/clock_recovery/clock_recovery_mm_fast.cpp), SDHIP_NDSP_MM_FAST, lane per (chunk, cadence): k_mmfast.
        // Returns the symbols written to d_out, or -1 when the call is to run as the one sequential lane (exact mode, a call too short for two chunks, lanes that
        // did not hand off). The hand-off is STRICT: chunk k's variant v stands if its state at the chunk start -- timing, rate, position, cadence counter, the
        // detector's delay lines -- is bit for bit the state its predecessor's standing variant ended with; so the output is the block's own, float for float.
        // A chunk none of whose variants fits runs again from that end state (up to 8 rounds); while a chunk is waiting for its re-run the scan goes on behind
        // it on the assumption that the re-run will end where one of its variants ended, and checks that afterwards.
        long long mmfast_stage(cf32 *A, long long n, float *d_out, size_t out_cap)
        {
            if (cfg.exact)
                return -1;
            const double gmu = std::max(1e-4f, cfg.clock_gain_mu);
            // merging bit for bit takes the RATE state down to its last bit: measured on the reference block, 12 - 25 k symbols at the default gains (8.7e-3), 4 - 11 k at
            // 0.02 -- 110 - 220 / gain_mu: 300 / gain_mu symbols of warm-up
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)(300.0 / gmu * final_sps);
            W = (env_int("SDHIP_W_MMFAST", W) + 255) / 256 * 256;
            const long long lanes = std::max<long long>(64, env_int("SDHIP_LANES_MMFAST", 13056));
            long long L = cfg.chunk_len > 0 ? cfg.chunk_len : std::max<long long>(4096, (n + lanes - 1) / lanes);
            L = (L + 63) / 64 * 64;
            if (L > (1 << 30) || W > (1 << 30))
                return -1;
            const ChunkGeom g = make_geom(n, (int)L, (int)W);
            if (g.K < 2)
                return -1;
            const int K = g.K;
            const double omin = (double)mm_p.omega_mid - std::fabs((double)mm_p.omega_limit);
            const int cap = ((int)((double)L / std::max(0.5, omin - 0.01)) + 16 + 7) & ~7;
            const int cap0 = ((int)((double)(L + W) / std::max(0.5, omin - 0.01)) + 16 + 7) & ~7;
            d_mf_rows.reserve((size_t)K * 6 * (size_t)cap + (size_t)cap0);
            d_mf_spec.reserve((size_t)K * 6);
            d_mf_end.reserve((size_t)K * 6);
            d_mf_counts.reserve((size_t)K * 6);
            d_mf_sel.reserve(K);
            d_mf_offs.reserve(K);
            d_mf_redo.reserve(K);
            d_mf_redo_start.reserve(K);
            put_hist(A, hist_cos);
            SD_HIP(hipMemcpyAsync(d_mm_start.p, &mm_s, sizeof(mm_s), hipMemcpyHostToDevice, stream));
            SD_HIP(hipMemsetAsync(d_mf_counts.p, 0, (size_t)K * 6 * sizeof(int), stream));
            launch_mmfast(A, d_mf_rows.p, d_mf_counts.p, g, cap, cap0, mm_p, d_mm_start.p, d_mf_spec.p, d_mf_end.p, nullptr, nullptr, 0, stream);
            std::vector<MmState> sp((size_t)K * 6), en((size_t)K * 6);
            std::vector<int> cnt((size_t)K * 6);
            SD_HIP(hipMemcpyAsync(sp.data(), d_mf_spec.p, sp.size() * sizeof(MmState), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(en.data(), d_mf_end.p, en.size() * sizeof(MmState), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(cnt.data(), d_mf_counts.p, cnt.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            auto same = [](const MmState &a, const MmState &b) {
                return memcmp(&a.mu, &b.mu, sizeof(float)) == 0 && memcmp(&a.omega, &b.omega, sizeof(float)) == 0 && a.inc == b.inc && a.upd_cnt == b.upd_cnt &&
                       memcmp(&a.p_1T, &b.p_1T, sizeof(cf32)) == 0 && memcmp(&a.p_0T, &b.p_0T, sizeof(cf32)) == 0 && memcmp(&a.c_1T, &b.c_1T, sizeof(cf32)) == 0 &&
                       memcmp(&a.c_0T, &b.c_0T, sizeof(cf32)) == 0;
            };
            std::vector<int> sel((size_t)K, 0);
            std::vector<char> exact_end((size_t)K, 0); // chunk k's standing row ends in a state that is known (not assumed)
            std::vector<int> assumed((size_t)K, -1);   // a chunk waiting for its re-run: the variant whose end state the scan behind it assumed
            std::vector<MmState> rr_start((size_t)K);  // the state a chunk's re-run (slot 5) started from ...
            std::vector<char> rr_done((size_t)K, 0);   // ... once it has run: it stands for as long as the predecessor still ends in that state
            long long fixed = 0;
            int rounds = 0;
            bool give_up = false;
            int from = 1; // chunks in front of `from` stand
            exact_end[0] = 1;
            for (;;)
            {
                std::vector<int> redo;
                std::vector<MmState> redo_start;
                // scan: the state chunk k has to start from = the end state of chunk k - 1's standing variant
                for (int k = from; k < K; k++)
                {
                    const MmState *prev = nullptr;
                    if (assumed[(size_t)k - 1] >= 0)
                        prev = &en[((size_t)k - 1) * 6 + (size_t)assumed[(size_t)k - 1]];
                    else
                        prev = &en[((size_t)k - 1) * 6 + (size_t)sel[(size_t)k - 1]];
                    int found = -1;
                    for (int v = 0; v < 5 && found < 0; v++)
                        if (same(sp[(size_t)k * 6 + v], *prev))
                            found = v;
                    if (found >= 0)
                    {
                        sel[(size_t)k] = found;
                        assumed[(size_t)k] = -1;
                        continue;
                    }
                    if (rr_done[(size_t)k] && assumed[(size_t)k - 1] < 0 && same(rr_start[(size_t)k], *prev))
                    { // re-run in an earlier round from exactly this state: its row and end state (slot 5) stand
                        sel[(size_t)k] = 5;
                        assumed[(size_t)k] = -1;
                        continue;
                    }
                    // none fits: re-run from the predecessor's end state if that is known; the scan goes on behind this chunk assuming its re-run will end where the
                    // variant of the predecessor's cadence ended (the lanes of one cadence have usually merged by then)
                    if (assumed[(size_t)k - 1] < 0)
                    {
                        redo.push_back(k);
                        redo_start.push_back(*prev);
                        rr_start[(size_t)k] = *prev;
                        rr_done[(size_t)k] = 0;
                    }
                    int guess = -1;
                    for (int v = 0; v < 5 && guess < 0; v++)
                        if (k + 1 < K)
                            for (int w = 0; w < 5 && guess < 0; w++)
                                if (same(sp[((size_t)k + 1) * 6 + w], en[(size_t)k * 6 + v]))
                                    guess = v;
                    assumed[(size_t)k] = guess >= 0 ? guess : 0;
                    sel[(size_t)k] = 5; // (re-run lanes fill slot 5)
                }
                if (redo.empty())
                    break;
                // (a stretch of stream on which the lanes need longer than the warm-up to merge fails as a run of neighbouring chunks: one round per chunk of it)
                if (++rounds > 40 || (long long)redo.size() * 4 > (long long)K)
                {
                    give_up = true;
                    break;
                }
                fixed += (long long)redo.size();
                SD_HIP(hipMemcpyAsync(d_mf_redo.p, redo.data(), redo.size() * sizeof(int), hipMemcpyHostToDevice, stream));
                SD_HIP(hipMemcpyAsync(d_mf_redo_start.p, redo_start.data(), redo_start.size() * sizeof(MmState), hipMemcpyHostToDevice, stream));
                launch_mmfast(A, d_mf_rows.p, d_mf_counts.p, g, cap, cap0, mm_p, d_mm_start.p, d_mf_spec.p, d_mf_end.p, d_mf_redo.p, d_mf_redo_start.p, (int)redo.size(), stream);
                int first = K;
                for (int k : redo)
                {
                    MmState ne;
                    SD_HIP(hipMemcpyAsync(&ne, d_mf_end.p + (size_t)k * 6 + 5, sizeof(MmState), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipMemcpyAsync(&cnt[(size_t)k * 6 + 5], d_mf_counts.p + (size_t)k * 6 + 5, sizeof(int), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    rr_done[(size_t)k] = 1;
                    const bool as_assumed = same(ne, en[(size_t)k * 6 + (size_t)assumed[(size_t)k]]);
                    if (getenv("SDHIP_DEBUG") && k < 64)
                        fprintf(stderr, "[sdhip]   mmfast chunk %d re-run: assumed variant %d, ended as assumed: %d (inc %lld mu %.9g omega %.9g cnt %u | assumed inc %lld mu %.9g omega %.9g cnt %u)\n", k,
                                assumed[(size_t)k], (int)as_assumed, ne.inc, ne.mu, ne.omega, ne.upd_cnt, en[(size_t)k * 6 + (size_t)assumed[(size_t)k]].inc,
                                en[(size_t)k * 6 + (size_t)assumed[(size_t)k]].mu, en[(size_t)k * 6 + (size_t)assumed[(size_t)k]].omega, en[(size_t)k * 6 + (size_t)assumed[(size_t)k]].upd_cnt);
                    en[(size_t)k * 6 + 5] = ne;
                    sel[(size_t)k] = 5;
                    assumed[(size_t)k] = -1;
                    if (!as_assumed)
                        first = std::min(first, k + 1); // what stood behind it stood on an assumption that did not hold: scanned again
                }
                // chunks that waited behind an assumed end state are scanned again from the first of them
                for (int k = 1; k < K; k++)
                    if (assumed[(size_t)k] >= 0)
                    {
                        first = std::min(first, k);
                        break;
                    }
                if (first >= K)
                    break;
                from = first;
                for (int k = from; k < K; k++)
                    if (assumed[(size_t)k] >= 0)
                        assumed[(size_t)k] = -1;
            }
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] mmfast chunks %d x 5 cadences, L %d W %d  re-run %lld in %d round(s)%s\n", K, g.L, g.W, fixed, rounds, give_up ? "  -> one sequential lane" : "");
            if (give_up)
            {
                stats.chunks_forced += 1;
                return -1;
            }
            std::vector<long long> offs((size_t)K);
            long long tot = 0;
            for (int k = 0; k < K; k++)
            {
                const int c = cnt[(size_t)k * 6 + (size_t)sel[(size_t)k]];
                if (c < 0)
                    throw HipError("fast_clock_recovery_mm_cc: symbol row overflow");
                offs[(size_t)k] = tot;
                tot += c;
            }
            if ((size_t)tot > out_cap)
                throw HipError("symbol output buffer too small");
            SD_HIP(hipMemcpyAsync(d_mf_sel.p, sel.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, stream));
            SD_HIP(hipMemcpyAsync(d_mf_offs.p, offs.data(), (size_t)K * sizeof(long long), hipMemcpyHostToDevice, stream));
            launch_mmfast_gather(d_mf_rows.p, d_mf_sel.p, d_mf_offs.p, d_mf_counts.p, K, cap, reinterpret_cast<cf32 *>(d_out), stream);
            get_hist(A, n, hist_cos);
            SD_HIP(hipStreamSynchronize(stream));
            mm_s = en[((size_t)K - 1) * 6 + (size_t)sel[(size_t)K - 1]];
            mm_s.inc -= n; // clock_recovery_mm_fast.cpp:153-156
            if (mm_s.inc < 0)
                mm_s.inc = 0;
            stats.chunks += K;
            stats.chunks_fixed += fixed;
            last_symbols = tot;
            return tot;
        }

        // ---- ndsp::CostasFastBlock (dsp/pll/costas_fast.cpp), SDHIP_NDSP_COSTAS_FAST. exact: one sequential lane, bit for bit. Otherwise lane-per-chunk like the
        // plain loop: warm-up lanes start from the carried frequency (first call: the M-th power estimate) next to a stable point, renorm_ctr follows the
        // stream's sample count and is therefore known at every chunk start; the hand-off is judged on the HOST from the K start / end states (a few MB): the
        // start state bit-identical to the predecessor's end state modulo an exact quarter / half turn (strict, the default: the output is then the block's own,
        // float for float) or inside the plain loop's windows (SDHIP_CF_STRICT=0). A chunk that
        // fails runs again from its predecessor's end state (and inherits its frame); the frames are turned back on the output (exact quarter / half turns).
        // A call whose lanes do not hand off (a loop that is not locked: more than an eighth of the boundaries fail, or the re-runs do not settle) is run as
        // the one sequential lane instead -- so the worst case is the block's own trajectory at twice the sequential lane's time.
        void costas_fast_stage(const cf32 *A, cf32 *B, long long n)
        {
            if (n > (1ll << 30))
                throw HipError("costas_fast_cc: at most 2^30 samples per call");
            cf_p.ctr_base = (unsigned)(cf_total % 65);
            cf_s.margin = 3.0e38f; // (per call)
            auto sequential = [&]() {
                SD_HIP(hipMemcpyAsync(d_cf_state.p, &cf_s, sizeof(cf_s), hipMemcpyHostToDevice, stream));
                launch_costas_fast(A, B, n, cf_p, d_cf_state.p, stream);
                SD_HIP(hipMemcpyAsync(&cf_s, d_cf_state.p, sizeof(cf_s), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                stats.chunks += 1;
            };
            const int L = pick_L(n, ST_COSTAS);
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)std::max(512.0, 24.0 / (1.414 * std::max(1e-5f, cfg.pll_bw)));
            W = std::max(W, w_cos_learned); // (a stream that needed a longer warm-up keeps it)
            W = (std::min<long long>(env_int("SDHIP_W_COSTAS", W), 1 << 20) + 255) / 256 * 256;
            ChunkGeom g = make_geom(n, L, (int)W);
            if (cfg.exact || g.K < 2)
            {
                sequential();
                cf_total += n;
                return;
            }
            cf_p.init_freq = started ? cf_s.freq : carrier_start_freq(A, n, 1.0);
            cf_p.init_freq = std::min(std::max(cf_p.init_freq, cf_p.fmin), cf_p.fmax);
            cf_p.est_len = (int)std::min<long long>(256, W / 2);
            int K = g.K;
            d_cf_spec.reserve(K);
            d_cf_end.reserve(K);
            d_cf_redo.reserve(K);
            d_rot.reserve(K);
            SD_HIP(hipMemcpyAsync(d_cf_state.p, &cf_s, sizeof(cf_s), hipMemcpyHostToDevice, stream));
            std::vector<CostasFastState> sp((size_t)K), en((size_t)K);
            auto first_pass = [&]() {
                launch_costas_fast_chunks(A, B, g, cf_p, d_cf_state.p, d_cf_spec.p, d_cf_end.p, nullptr, 0, stream);
                SD_HIP(hipMemcpyAsync(sp.data() + 1, d_cf_spec.p + 1, (size_t)(K - 1) * sizeof(CostasFastState), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipMemcpyAsync(en.data(), d_cf_end.p, (size_t)K * sizeof(CostasFastState), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
            };
            first_pass();
            const double tol_phase = env_int("SDHIP_COSTAS_TOL_URAD", 10000) * 1e-6, tol_freq = env_int("SDHIP_COSTAS_TOL_NFREQ", 40000) * 1e-9;
            // SDHIP_CF_STRICT (default 1): a hand-off stands only when it is bit-identical -- this loop's lanes DO merge with the sequential trajectory bit for bit
            // (no transcendental function in the recurrence; measured on the twin: every boundary behind the acquisition) --, so the whole output is the block's
            // own, float for float, computed lane-per-chunk. 0: the plain loop's tolerance windows.
            const bool strict = env_int("SDHIP_CF_STRICT", 1) != 0;
            std::vector<int> turn((size_t)K, 0), redo;
            std::vector<char> rerun((size_t)K, 0), window_ok((size_t)K, 1);
            long long outside = 0; // boundaries outside the tolerance windows: lanes that did not reach the sequential trajectory (what the warm-up is judged by)
            // boundary k: chunk k's start state against chunk k - 1's end state; returns whether it stands, turn[k] = its frame relative to its predecessor's
            const bool dbg = getenv("SDHIP_DEBUG") != nullptr;
            auto judge = [&](int k) {
                const CostasFastState &a = sp[(size_t)k], &b = en[(size_t)k - 1];
                if (strict && !dbg && a.fre_re == b.fre_re && a.fre_im == b.fre_im && a.ctr == b.ctr)
                { // the common case without a transcendental function: which exact turn, if any, carries the start phasor onto the predecessor's
                    float pr = a.pha_re, pi = a.pha_im;
                    for (int quarter = 0; quarter < 4; quarter++)
                    {
                        if (pr == b.pha_re && pi == b.pha_im && (order != 2 || (quarter & 1) == 0))
                        {
                            turn[(size_t)k] = order == 2 ? quarter >> 1 : (order == 4 ? quarter : 2 * quarter);
                            window_ok[(size_t)k] = 1;
                            return true;
                        }
                        const float tr = -pi;
                        pi = pr;
                        pr = tr;
                    }
                }
                const double pa = std::atan2(-(double)a.pha_im, (double)a.pha_re), pb = std::atan2(-(double)b.pha_im, (double)b.pha_re);
                const double r = std::floor((pa - pb) / rot_unit + 0.5), res = (pa - pb) - r * rot_unit;
                const double ma = std::hypot((double)a.pha_re, (double)a.pha_im), mb = std::hypot((double)b.pha_re, (double)b.pha_im);
                const double fa = std::hypot((double)a.fre_re, (double)a.fre_im), fb = std::hypot((double)b.fre_re, (double)b.fre_im);
                const double dfr = std::atan2(-(double)a.fre_im, (double)a.fre_re) - std::atan2(-(double)b.fre_im, (double)b.fre_re);
                turn[(size_t)k] = (int)(((long long)r % rot_mod + rot_mod) % rot_mod);
                if (dbg && k < 5)
                    fprintf(stderr, "[sdhip] costas_fast boundary %d: phase %.6f vs %.6f (res %.2e)  freq %.6f vs %.6f  arg(fre) diff %.2e  |pha| %.6f vs %.6f  |fre| %.6f vs %.6f  ctr %u vs %u\n", k, pa, pb,
                            res, a.freq, b.freq, dfr, ma, mb, fa, fb, a.ctr, b.ctr);
                const bool in_window = std::fabs(res) < tol_phase && std::fabs((double)a.freq - (double)b.freq) < tol_freq && std::fabs(dfr) < tol_freq && std::fabs(ma - mb) < 1e-4 &&
                                       std::fabs(fa - fb) < 1e-4 && a.ctr == b.ctr;
                window_ok[(size_t)k] = in_window ? 1 : 0;
                if (strict)
                { // the start state IS the predecessor's end state, a whole number of exact (quarter / half) turns on: every float of the chunk then equals the
                  // sequential lane's (the loop's arithmetic commutes with those turns bit for bit: products and sums only change sign or place)
                    const int q = turn[(size_t)k];
                    if (order == 8 && (q & 1))
                        return false;
                    const int quarter = order == 2 ? 2 * (q & 1) : (order == 4 ? q & 3 : (q >> 1) & 3);
                    float pr = a.pha_re, pi = a.pha_im;
                    for (int t = 0; t < quarter; t++)
                    {
                        const float tr = -pi;
                        pi = pr;
                        pr = tr;
                    }
                    return pr == b.pha_re && pi == b.pha_im && a.fre_re == b.fre_re && a.fre_im == b.fre_im && a.ctr == b.ctr; // (freq: see the limiter guard below)
                }
                return in_window;
            };
            auto judge_all = [&]() {
                redo.clear();
                outside = 0;
                if ((int)turn.size() < K)
                    throw HipError("costas_fast: chunk count grew");
                for (int k = 1; k < K; k++)
                {
                    if (!judge(k))
                        redo.push_back(k);
                    outside += window_ok[(size_t)k] ? 0 : 1;
                }
            };
            judge_all();
            // many lanes short of the sequential trajectory: the loop's gain goes with the signal's amplitude (no AGC inside the block), so the warm-up the nominal
            // bandwidth suggests may be short -- the lanes' median end frequency as the start value, then twice the warm-up, up to three times (the stream keeps it)
            for (int attempt = 0; attempt < 4 && outside * 8 > (long long)(K - 1); attempt++)
            {
                std::vector<float> fr((size_t)K);
                for (int k = 0; k < K; k++)
                    fr[(size_t)k] = en[(size_t)k].freq;
                std::nth_element(fr.begin(), fr.begin() + fr.size() / 2, fr.end());
                const float med = fr[fr.size() / 2];
                if (attempt == 0 && std::fabs(med - cf_p.init_freq) > 0.05f * cfg.pll_bw)
                    cf_p.init_freq = med;
                else
                {
                    const long long w2 = 2 * (long long)g.W;
                    const ChunkGeom g2 = make_geom(n, L, (int)w2);
                    if (cfg.warmup > 0 || getenv("SDHIP_W_COSTAS") || g2.K < 2 || w2 > (1 << 20))
                        break;
                    g = g2;
                    K = g.K;
                    w_cos_learned = w2;
                    cf_p.est_len = (int)std::min<long long>(256, w2 / 2);
                }
                first_pass();
                judge_all();
            }
            bool give_up = outside * 8 > (long long)(K - 1);
            int rounds = 0;
            long long fixed = 0;
            while (!give_up && !redo.empty())
            {
                if (++rounds > 16 || (rounds > 3 && (long long)redo.size() * 8 > (long long)(K - 1)))
                {
                    give_up = true;
                    break;
                }
                for (int k : redo)
                { // from the predecessor's end state, in the predecessor's frame
                    sp[(size_t)k] = en[(size_t)k - 1];
                    turn[(size_t)k] = 0;
                    rerun[(size_t)k] = 1;
                }
                fixed += (long long)redo.size();
                SD_HIP(hipMemcpyAsync(d_cf_spec.p, sp.data(), (size_t)K * sizeof(CostasFastState), hipMemcpyHostToDevice, stream));
                SD_HIP(hipMemcpyAsync(d_cf_redo.p, redo.data(), redo.size() * sizeof(int), hipMemcpyHostToDevice, stream));
                launch_costas_fast_chunks(A, B, g, cf_p, d_cf_state.p, d_cf_spec.p, d_cf_end.p, d_cf_redo.p, (int)redo.size(), stream);
                SD_HIP(hipMemcpyAsync(en.data(), d_cf_end.p, (size_t)K * sizeof(CostasFastState), hipMemcpyDeviceToHost, stream));
                SD_HIP(hipStreamSynchronize(stream));
                std::vector<int> next;
                for (int k : redo)
                    if (k + 1 < K && !judge(k + 1)) // (a chunk re-run in this very round started from its predecessor's OLD end state: judged again like any other)
                        next.push_back(k + 1);
                redo.swap(next);
            }
            // The limiter guard of the strict hand-off. `freq` only accumulates beta * error: a lane that merged in (pha, fre) carries from then on the sequential
            // lane's freq plus the offset it had at its chunk start (the sums' own rounding apart: at most half an ulp of the largest |freq| per sample, a random walk). The offsets
            // chain over the boundaries, so chunk k's freq is off by no more than D_k = |sum of (start freq - predecessor's end freq) up to k| + samples x ulp; its
            // limiter decisions are the sequential lane's if freq stayed further than D_k from both limits at every renormalisation it ran (margin; its warm-up's
            // included). A chunk that cannot show that sends the call to the one sequential lane.
            if (strict && !give_up)
            {
                double off = 0.0, fmaxabs = 0.0;
                for (int k = 0; k < K; k++)
                    fmaxabs = std::max(fmaxabs, (double)std::fabs(en[(size_t)k].freq));
                const double ulp = std::ldexp(1.0, std::ilogb(std::max(fmaxabs, 1e-30)) - 23);
                for (int k = 1; k < K && !give_up; k++)
                {
                    off += (double)sp[(size_t)k].freq - (double)en[(size_t)k - 1].freq;
                    const double ne = (double)chunk_end(g, k);
                    const double Dk = std::fabs(off) + std::min(ne, 16.0 * std::sqrt(ne)) * ulp + 1e-7; // (the sums' rounding walk: its worst case, or 16 standard deviations)
                    if (!((double)en[(size_t)k].margin > Dk))
                        give_up = true;
                }
                if (en[0].margin < 0.0f) // the sequential lane's own limiter acted: exact by itself, but what follows started from lanes that did not see it
                    give_up = true;
            }
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] costas_fast chunks %d L %d W %d  re-run %lld in %d round(s)%s\n", K, g.L, g.W, fixed, rounds, give_up ? "  -> one sequential lane" : "");
            if (give_up)
            {
                sequential();
                stats.chunks_forced += 1;
                cf_total += n;
                return;
            }
            // frames: rot[k] = sum of the turns up to k; the output of chunk k is out * exp(-j rot unit), the state pha * exp(-j rot unit): turned back by rot_apply
            std::vector<int> rot((size_t)K, 0);
            for (int k = 1; k < K; k++)
                rot[(size_t)k] = (rot[(size_t)k - 1] + turn[(size_t)k]) % rot_mod;
            SD_HIP(hipMemcpyAsync(d_rot.p, rot.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, stream));
            launch_derotate(B, n, g, d_rot.p, order, stream);
            cf_s = en[(size_t)K - 1];
            {
                const double a = rot[(size_t)K - 1] * rot_unit, c = std::cos(a), sn = std::sin(a);
                const int q = rot[(size_t)K - 1];
                double pr = cf_s.pha_re, pi = cf_s.pha_im;
                if (order == 2 || (order == 4) || (order == 8 && (q & 1) == 0))
                { // exact half / quarter turns
                    const int quarter = order == 2 ? 2 * (q & 1) : (order == 4 ? q & 3 : (q >> 1) & 3);
                    for (int t = 0; t < quarter; t++)
                    {
                        const double tr = -pi;
                        pi = pr;
                        pr = tr;
                    }
                }
                else
                {
                    const double tr = pr * c - pi * sn, ti = pr * sn + pi * c;
                    pr = tr;
                    pi = ti;
                }
                cf_s.pha_re = (float)pr;
                cf_s.pha_im = (float)pi;
            }
            SD_HIP(hipStreamSynchronize(stream));
            stats.chunks += K;
            stats.chunks_fixed += fixed;
            cf_total += n;
        }

        void costas_stage(const cf32 *A, cf32 *B, long long n, ChunkGeom &cg, double sps, double rate_hz)
        {
            if (!started)
                cos_p.init_freq = carrier_start_freq(A, n, sps);
            else
                cos_p.init_freq = cos_s.freq;
            // both loop modes decay like exp(-zeta*wn*t) with zeta*wn ~ 1.414*pll_bw per sample (unit detector gain after
            // the AGC): 24 time constants from a phase error of up to pi/order bring the warm-up to the float floor of two
            // trajectories of this loop on the same samples. A stream that needed more (first judgement failed widely, see
            // the respec hook below) keeps the longer warm-up for its later calls.
            const long long w_cos_cap = 1 << 20;
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)std::max(512.0, 24.0 / (1.414 * std::max(1e-5f, cfg.pll_bw)));
            W = std::max(W, w_cos_learned);
            W = env_int("SDHIP_W_COSTAS", W);
            W = (std::min<long long>(W, w_cos_cap) + 255) / 256 * 256;
            const int L = pick_L(n, ST_COSTAS);
            // Acceptance window of a Costas boundary = the soft-symbol parity target (1e-5 relative): a phase offset of d rad
            // is a relative symbol error of d. Two trajectories of this loop on the same samples contract onto each other down to
            // float noise (measured, tools/twin/soft_parity.py and DESIGN.md 2: median 5e-7, p99 3e-6 rad) except while one of the
            // sign detectors of the order-4/8 error has just disagreed (a kick of ~alpha that decays within a few hundred
            // samples): such boundaries fail the window and their chunk is re-run from the exact state until it has merged.
            // RE-RUN window: 1e-2 rad / 4e-5 rad/sample. Boundaries outside 1e-5 rad are the ones where one of the two
            // trajectories was kicked shortly before the boundary (sign detectors disagreeing, see above): measured on MetOp,
            // 196 k boundaries per 16 GiB step: 71 beyond 1e-5 rad, 54 beyond 1e-4. Each decays under 1e-5 within
            // ~tau ln(d / 1e-5) samples (tau ~ 230: <= 1600 samples of a chunk of >= 10^4 at d = 1e-2), so letting them stand
            // costs ~5e-5 of the symbols a transient below the loop's own phase jitter, while re-running them costs a second
            // launch whose slowest lane runs alone for over a millisecond (measured: +1.2 ms on a 96 ms step). They are counted
            // (chunks_inexact); anything beyond the window -- a lane that has not locked -- is re-run from the exact state.
            const double tol_phase = env_int("SDHIP_COSTAS_TOL_URAD", 10000) * 1e-6, tol_freq = env_int("SDHIP_COSTAS_TOL_NFREQ", 40000) * 1e-9;
            // Round 6 measured exactly that again, with the re-run lanes stopping at a checkpoint (round 4): a TIGHT window (SDHIP_COSTAS_TIGHT_URAD, in 1e-6 rad;
            // boundaries between it and the wide window re-run from the exact state until they are back within it; the wide window keeps its role for the warm-up
            // adaptation, VerdictOut::wide). At 1e-5 rad, full size (profiles/r06_b_*): GOES 6 084 of 295 k chunks re-run, same-arm symbols beyond 1e-5 6 783 ->
            // 2 445 of 12.4 M; NPP 1 485 re-run, 16 536 -> 13 013 of 20 M; MetOp 66 re-run, 4 721 -> 5 164 (nothing: what is left there are sign-detector
            // disagreements in MID-chunk, tools/twin/arm_probe.py) -- for +1.2 ms per step on every workload (the slowest re-run lane walks its 2048
            // samples alone: 0.58 us per sample of dependent chain), +3 % on MetOp and +9 % on GOES for a change in the fifth digit of the 1e-5 fraction. So the
            // default stays the wide window; the switch is there for a caller who wants the hand-offs inside the contract and pays for it.
            const double tol_tight = std::min(tol_phase, env_int("SDHIP_COSTAS_TIGHT_URAD", 10000) * 1e-6);
            ChunkCkpt cos_ck;
            auto costas_setup = [&](long long Wn) {
                cos_p.est_len = (int)std::min<long long>(env_int("SDHIP_COSTAS_EST", 256), Wn / 2);
                cg = make_geom(n, L, (int)Wn);
                d_cos_spec.reserve(cg.K);
                d_cos_end.reserve(cg.K);
                if (use_ckpt)
                {
                    cos_ck.len = 2048;
                    cos_ck.per_chunk = L / cos_ck.len + 1;
                    d_cos_ck.reserve((size_t)cg.K * cos_ck.per_chunk);
                    cos_ck.ck = d_cos_ck.p;
                    cos_ck.tol_a = (float)tol_tight;
                    cos_ck.tol_b = (float)tol_freq;
                    if (getenv("SDHIP_DEBUG"))
                    {
                        d_ck_work.reserve(6);
                        cos_ck.work = d_ck_work.p + 3;
                    }
                }
                d_rot.reserve(cg.K);
                d_dm.reserve(cg.K);
            };
            costas_setup(W);
            SD_HIP(hipMemcpyAsync(d_cos_start.p, &cos_s, sizeof(cos_s), hipMemcpyHostToDevice, stream));
            launch_costas(A, B, cg, cos_p, d_cos_start.p, d_cos_spec.p, d_cos_end.p, nullptr, 0, stream, cos_ck);
            verify_fix(
                "costas", cg.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_costas_verdict, dim3((cg.K + 255) / 256), dim3(256), 0, stream, cg.K, d_cos_spec.p, d_cos_end.p, rot_unit, rot_mod, tol_phase, tol_tight, tol_freq,
                                       d_dm.p, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    if (use_ckpt)
                        hipLaunchKernelGGL(k_costas_spec_aligned, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_cos_spec.p, d_cos_end.p, rot_unit);
                    else
                        hipLaunchKernelGGL(k_spec_from_prev<CostasState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_cos_spec.p, d_cos_end.p);
                },
                [&](const int *redo, int nr) { launch_costas(A, B, cg, cos_p, d_cos_start.p, d_cos_spec.p, d_cos_end.p, redo, nr, stream, cos_ck); },
                [&](int) {
                    // Many warm-ups missed. (a) The start frequency was off (the M-th-power estimate is weak for order 8 and at low
                    // SNR; a call may also begin in noise with the carried loop state meaningless): every lane has meanwhile run a
                    // real loop over W + L samples, and the median of their end frequencies is a far better start value. (b) The
                    // warm-up is too short for this signal's loop dynamics: double it (the stream keeps the longer one).
                    std::vector<CostasState> es((size_t)cg.K);
                    SD_HIP(hipMemcpyAsync(es.data(), d_cos_end.p, es.size() * sizeof(CostasState), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    std::vector<float> fr(es.size());
                    for (size_t i = 0; i < es.size(); i++)
                        fr[i] = es[i].freq;
                    std::nth_element(fr.begin(), fr.begin() + fr.size() / 2, fr.end());
                    const float med = fr[fr.size() / 2];
                    if (std::fabs(med - cos_p.init_freq) > 0.05f * cfg.pll_bw)
                        cos_p.init_freq = med;
                    else if (cfg.warmup <= 0 && !getenv("SDHIP_W_COSTAS") && 2 * (long long)cg.W <= w_cos_cap)
                    {
                        w_cos_learned = 2 * (long long)cg.W;
                        costas_setup(w_cos_learned);
                    }
                    else
                        return false;
                    launch_costas(A, B, cg, cos_p, d_cos_start.p, d_cos_spec.p, d_cos_end.p, nullptr, 0, stream, cos_ck);
                    return true;
                },
                true);
            stats.chunks += cg.K;
            if (cos_ck.work)
                ck_report("costas", 1);
            // rot[k] = frame of chunk k relative to the stream's (prefix sum of the per-boundary turns)
            {
                const int nt = (cg.K + 1023) / 1024;
                d_tile_sums.reserve(nt);
                hipLaunchKernelGGL(k_chunk_scan_sums, dim3(nt), dim3(1024), 0, stream, cg.K, 0, d_dm.p, nullptr, nullptr, nullptr, 0, d_tile_sums.p, nullptr);
                hipLaunchKernelGGL(k_chunk_scan_apply, dim3(nt), dim3(1024), 0, stream, cg.K, 0, d_dm.p, rot_mod, d_rot.p, nullptr, nullptr, nullptr, d_tile_sums.p,
                                   nullptr, nullptr, nullptr);
            }
            int rot_last = 0;
            SD_HIP(hipMemcpyAsync(&cos_s, d_cos_end.p + (cg.K - 1), sizeof(cos_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(&rot_last, d_rot.p + (cg.K - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            // chunk k's phase = true phase + rot[k]*unit; carry the loop state in the frame of the last chunk,
            // re-expressed in the stream's frame (rot 0) so the next call starts unrotated
            if (rot_last != 0)
            {
                double ph = (double)cos_s.phase - rot_last * rot_unit;
                while (ph > 2 * design::PI)
                    ph -= 2 * design::PI;
                while (ph < -2 * design::PI)
                    ph += 2 * design::PI;
                cos_s.phase = (float)ph;
            }
            stats.freq_hz = (float)(((double)cos_s.freq / (2.0 * design::PI)) * rate_hz);
        }

        // ---- M&M + quantiser: A (history in front) -> d_soft / d_syms; returns the soft symbols written (last_symbols: the symbols)
        long long last_symbols = 0;
        int64_t mm_stage(cf32 *A, long long n, const ChunkGeom &cg, const int *mm_rot, int8_t *d_soft, size_t soft_cap, float *d_syms, size_t syms_cap,
                         std::vector<cf32> &hist)
        {
            put_hist(A, hist);
            // timing loop: ~2/(Kd*gain_mu) symbols per time constant with a detector gain Kd well below 1 at low Es/N0
            // (measured: ~700 symbols at 7 dB BPSK with the default gains)
            // gear-shifted warm-up (tools/mm_gear_study.py): ~2.75/gain_mu symbols at 8x the timing gain (rate term frozen) pull the
            // phase in, ~16/gain_mu symbols at the loop's own gains settle it onto the sequential trajectory
            // (|dt| p99 < 0.03 sample at GOES' 7 dB, far less for the QPSK configs); 36/gain_mu without the fast gear
            const double gmu = std::max(1e-4f, cfg.clock_gain_mu);
            // When nobody asks for the float symbols (the modules, bench.py's timed steps) the clock recovery stores the module's int8 soft symbols itself -- two bytes
            // per symbol in its scratch rows instead of eight, eight symbols per 16-byte store -- and the compaction behind it moves a quarter of the bytes
            // (k_compact8 in place of k_quantize). Round 2 had measured this slower (k_mm was bound by its instruction issue then: +1.3 - 2.2 ms for ~15 more
            // instructions per symbol); on round 5's k_mm, which is not, it is faster (visit L: dword-paired stores and a halfword-wise compaction already level with
            // the float rows). Same bytes either way (test_soft_symbols_without_the_float_symbols). SDHIP_MM_Q8=0: float rows + k_quantize.
            // (The Gardner lanes and the test tap carry float symbols only. BPSK keeps the float rows by default: half of an int8 pair is dropped again by the
            // compaction, and on GOES -- five sixths of the lanes' work are warm-up, where nothing is stored -- the int8 instance measured 0.3 ms behind, visits M / N.)
            mm_p.q8 = (d_syms == nullptr && tap_mode == 0 && mm_p.loop != 1 && env_int("SDHIP_MM_Q8", is_bpsk ? 0 : 1) != 0) ? 1 : 0;
            mm_p.q8_bpsk = is_bpsk ? 1 : 0;
            mm_p.tap = tap_mode;
            mm_p.fast = (!cfg.exact && mm_p.loop != 2 && env_int("SDHIP_FAST_MATH", 1) != 0) ? 1 : 0;
            mm_p.fast_mult = (float)env_int("SDHIP_MM_FAST_MULT", 8);
            mm_p.fast_syms = mm_p.fast_mult > 1.0f ? (int)env_int("SDHIP_MM_FAST_SYMS", (long long)(2.75 / gmu)) : 0;
            // a re-run costs one lane the whole chunk, so long chunks (large batches) keep the conservative warm-up: it is a
            // small fraction of L there anyway
            const int L = pick_L(n, ST_MM);
            // Gardner's detector is the product of two interpolated samples: its gain goes with the signal power (0.36 behind an AGC at 0.6) where the
            // M&M detector's goes with the amplitude, so the same loop gains give a time constant ~4 times as long (measured on the twin: QPSK at 10 dB,
            // muGain 8.7e-3: 1.5 % of the symbols beyond 1e-5 with 22 k symbols of warm-up, 36 % with 7 k)
            const double slow = mm_p.loop == 1 ? 4.0 : 1.0;
            const double w_full = slow * 36.0 / gmu * final_sps;
            const double w_gear = mm_p.fast_syms > 0 ? (mm_p.fast_syms + slow * 16.0 / gmu) * final_sps : w_full;
            // Warm-up length. The timing loop's contraction rate depends on the detector gain, i.e. on the signal (measured time
            // constants: ~360 symbols for MetOp QPSK at 10 dB, ~870 for GOES BPSK at 7 dB; tools/twin/soft_parity.py),
            // and the chunk's symbols only agree with the sequential reference's once the warm-up has brought the lane within
            // ~1e-4 sample of its trajectory. So the first guess (gear-shifted ~19/gain_mu symbols) is checked against the tight
            // hand-off window below, and if more than an eighth of the boundaries miss it the stage is launched again with twice
            // the warm-up (up to 64 loop constants 1/gain_mu); the stream keeps what it learned for its later calls.
            // A caller's own hand-off windows (demod_set_mm_windows: the DVB-S2 module, see MM_TOL_TIGHT below) hold from the stream's SECOND call on: while the
            // loops are still pulling the signal in (the first call of a stream) lanes are only comparable to the one sequential trajectory inside the default windows
            // -- with the wide ones the module lost the two earliest frames the reference decodes (visit I of round 5). Each pair of windows learns its own warm-up.
            const bool s2_front = mm_windows_tight > 0.0 && !cfg.exact && mm_calls > 0;
            mm_calls++;
            long long &w_learned = s2_front ? w_mm_learned_own : w_mm_learned;
            const long long w_mm_cap = (long long)(slow * 64.0 / gmu * final_sps);
            long long W = cfg.warmup > 0 ? cfg.warmup : (long long)std::min(w_full, std::max(w_gear, 0.5 * L));
            W = std::max(W, w_learned);
            W = env_int("SDHIP_W_MM", W);
            W = (W + 255) / 256 * 256;
            ChunkGeom g;
            // Hand-off windows of an M&M boundary, in samples of timing. Two trajectories of this loop on the same samples
            // hover 3e-5 ... 3e-4 sample apart (the feedback is piecewise constant in mu through the arm index), which makes
            // them pick different interpolator arms on 0.3-0.7 % of the symbols -- the floor of any time-parallel schedule.
            //  * TIGHT (2e-4): a boundary inside it adds nothing to that floor. It is the yardstick of the warm-up length: when more
            //    than an eighth of the boundaries miss it, the warm-up is doubled (respec below).
            //  * RE-RUN (5e-3): a boundary outside it is re-run from the exact state (and stops as soon as it is back inside). Between
            //    the two windows a chunk starts <= 5e-3 sample off and is on the floor again within a loop time constant (a few
            //    hundred symbols of a chunk of thousands): measured on MetOp, 14 of 65 k boundaries per step lie between 1e-3 and
            //    the re-run window; re-running them moves the 1e-5 fraction in the sixth digit and costs a second launch whose
            //    slowest lane runs alone for milliseconds.
            // The DVB-S2 front end (nd.skip_costas: sdhip_dvbs2_front_create) has its own pair of windows. Its clock recovery runs at the module's gain 1.7e-3 on a
            // roll-off of 0.2: two trajectories of THAT loop hover ~1e-2 sample apart for good (measured, visit G of round 5: 8 500 of 43 k boundaries outside 5e-3
            // behind 44 k samples of warm-up, dt 5e-3 .. 1.5e-2 behind 75 k), so MetOp's windows drove its adaptive warm-up to the cap -- 75 520 samples in front of
            // 2 048-sample chunks, 97 % of the lanes' work -- for nothing its consumers see: what is promised there is the decoders' output (the same BBFRAMEs, DESIGN
            // 4b), and a symbol taken 2e-2 sample off is 46 dB below the symbol. Windows of 2.5 / 5 interpolator arms instead.
            // (Set by the DVB-S2 demodulator MODULE's handle on its front end, demod_set_mm_windows: the front end as a unit -- sdhip_dvbs2_front_create -- keeps MetOp's.)
            const double MM_TOL_TIGHT = s2_front ? mm_windows_tight : 2e-4;
            const double MM_TOL = getenv("SDHIP_MM_TOL_MICRO") ? env_int("SDHIP_MM_TOL_MICRO", 5000) * 1e-6
                                  : (getenv("SDHIP_MM_TOL_MILLI") ? env_int("SDHIP_MM_TOL_MILLI", 5) * 1e-3 : (s2_front ? mm_windows_tol : 5e-3));
            // ... and the window of the loop's RATE state: 1e-3 of omega. Round 6 tried the window that would keep a rate error's excursion inside the timing
            // window (MM_TOL gain_mu / 2 samples per symbol: a lane whose timing passes through the predecessor's while its rate is still off is carried away again
            // before the critically damped loop pulls it back): GOES full size, symbols two or more arms from the reference's position 5 812 -> 5 017 of 12.4 M for
            // +1.0 ms of re-run launches per step (profiles/r06_b_*). Those runs of symbols are not hand-offs: they begin in MID-chunk, where the detector's slicer
            // (clock_recovery_mm.cpp:99-100) decided a near-zero symbol differently on the two trajectories -- a kick of ~gain_mu, one arm, on top of the flicker
            // (tools/twin/arm_probe.py). SDHIP_MM_TOL_OMEGA_NANO sets the window in 1e-9 samples per symbol (experiments).
            const float MM_TOL_OMEGA = getenv("SDHIP_MM_TOL_OMEGA_NANO") ? (float)(env_int("SDHIP_MM_TOL_OMEGA_NANO", 0) * 1e-9) : 1e-3f * final_sps;
            mm_p.tol_omega = MM_TOL_OMEGA;
            MmCkpt *ckp = nullptr;
            int ck_per_chunk = 0;
            auto mm_setup = [&](long long Wn) {
                g = make_geom(n, L, (int)Wn);
                const double omin = (double)mm_p.omega_mid - (double)mm_p.omega_limit;
                const long long span0 = std::min<long long>(n, (long long)L + Wn);
                mm_p.cap = ((int)(span0 / std::max(0.5, omin - 0.01)) + 16 + 7) & ~7; // whole groups of eight: int8 rows start on 16 bytes (k_mm<.., Q8>'s group stores)
                mm_p.cg = cg;
                mm_p.rot = mm_rot;
                symbuf.reserve((size_t)g.K * mm_p.cap);
                d_counts.reserve(2 * (size_t)g.K);
                d_offsets.reserve(g.K);
                d_mm_spec.reserve(g.K);
                d_mm_end.reserve(g.K);
                d_mm_spec_c.reserve(g.K);
                d_mm_end_c.reserve(g.K);
                d_skip.reserve(g.K);
                d_extra.reserve(g.K);
                d_seg.reserve(2 * (size_t)g.K);
                ck_per_chunk = L / MM_CK_SAMPLES + 2;
                if (use_ckpt || env_int("SDHIP_MM_CKPT", 0))
                { // checkpoints for the early exit of re-run lanes, k_mm<true>
                    d_mm_ck.reserve((size_t)g.K * ck_per_chunk);
                    ckp = d_mm_ck.p;
                }
            };
            mm_setup(W);
            if (getenv("SDHIP_PRINT_ADDR")) // experiment: the M&M launch time against where its buffers lie
                fprintf(stderr, "[sdhip] mm buffers: in %p  symbols %p (%zu B, row %d B)  K %d L %d W %d\n", (const void *)A, (void *)symbuf.p, symbuf.cap * sizeof(cf32),
                        mm_p.cap * 8, g.K, g.L, g.W);
            SD_HIP(hipMemcpyAsync(d_mm_start.p, &mm_s, sizeof(mm_s), hipMemcpyHostToDevice, stream));
            launch_mm(A, symbuf.p, d_counts.p, g, mm_p, d_mm_start.p, d_mm_spec.p, d_mm_end.p, d_mm_spec_c.p, d_mm_end_c.p, nullptr, 0, stream, ckp, ck_per_chunk,
                      (float)MM_TOL);
            // Symbol hand-off at chunk boundaries. The M&M loop never re-merges bit for bit: its feedback is piecewise
            // constant through the 128-arm interpolator index rint(mu*128) (clock_recovery_mm.cpp:66), so independent
            // trajectories hover a fraction of an arm apart (tools/merge_study.py). What is certified is CONSISTENCY in
            // time: t = inc + mu of the first symbol of chunk k (from its own warm-up) against the next-symbol time chunk
            // k-1 ended with. Equal within MM_TOL samples: chunk k stands. Exactly one or two symbol periods apart (the
            // boundary fell between the two trajectories' sample indices, mu wrapping on opposite sides): the symbol(s) are
            // taken from chunk k-1's look-ahead, or skipped at the head of chunk k. Anything else: re-run from the exact state.
            // (k_mm_verdict; the compaction segments and offsets are a prefix sum on the device, k_chunk_scan.)
            verify_fix(
                "mm", g.K,
                [&](VerdictOut *vo, int *fails, int force) {
                    hipLaunchKernelGGL(k_mm_verdict, dim3((g.K + 255) / 256), dim3(256), 0, stream, g.K, d_mm_spec_c.p, d_mm_end_c.p, d_counts.p, MM_TOL, std::min(MM_TOL, MM_TOL_TIGHT),
                                       MM_TOL_OMEGA, d_skip.p, d_extra.p, vo, fails, force);
                },
                [&](const int *list, int nr) {
                    if (getenv("SDHIP_DEBUG"))
                    { // first few rejected boundaries of the round
                        int idx[4];
                        const int m = std::min(nr, 4);
                        SD_HIP(hipMemcpy(idx, list, m * sizeof(int), hipMemcpyDeviceToHost));
                        for (int q = 0; q < m; q++)
                        {
                            MmCert a, b;
                            SD_HIP(hipMemcpy(&a, d_mm_spec_c.p + idx[q], sizeof(a), hipMemcpyDeviceToHost));
                            SD_HIP(hipMemcpy(&b, d_mm_end_c.p + idx[q] - 1, sizeof(b), hipMemcpyDeviceToHost));
                            const double d = (double)(a.inc - b.inc) + ((double)a.mu - (double)b.mu);
                            fprintf(stderr, "[sdhip] mm boundary %d rejected: dt %.5f samples = %.3f symbols, omega %.6f vs %.6f\n", idx[q], d, d / b.omega, a.omega, b.omega);
                        }
                    }
                    hipLaunchKernelGGL(k_spec_from_prev<MmCert>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_mm_spec_c.p, d_mm_end_c.p);
                    hipLaunchKernelGGL(k_spec_from_prev<MmState>, dim3((nr + 255) / 256), dim3(256), 0, stream, list, nr, d_mm_spec.p, d_mm_end.p);
                },
                [&](const int *redo, int nr) {
                    launch_mm(A, symbuf.p, d_counts.p, g, mm_p, d_mm_start.p, d_mm_spec.p, d_mm_end.p, d_mm_spec_c.p, d_mm_end_c.p, redo, nr, stream, ckp,
                              ck_per_chunk, (float)MM_TOL);
                },
                [&](int nf) {
                    const long long wn = (std::min<long long>(2 * (long long)g.W, w_mm_cap) + 255) / 256 * 256;
                    if (cfg.warmup > 0 || getenv("SDHIP_W_MM") || wn <= (long long)g.W)
                        return false;
                    w_learned = wn;
                    if (getenv("SDHIP_DEBUG"))
                        fprintf(stderr, "[sdhip] mm     %d of %d boundaries outside the hand-off window: warm-up %d -> %lld samples\n", nf, g.K, g.W, w_learned);
                    mm_setup(w_learned);
                    launch_mm(A, symbuf.p, d_counts.p, g, mm_p, d_mm_start.p, d_mm_spec.p, d_mm_end.p, d_mm_spec_c.p, d_mm_end_c.p, nullptr, 0, stream, ckp,
                              ck_per_chunk, (float)MM_TOL);
                    return true;
                });
            stats.chunks += g.K;
            // compaction segments + offsets + total
            SD_HIP(hipMemsetAsync(d_vout.p, 0, sizeof(VerdictOut), stream));
            {
                const int nt = (g.K + 1023) / 1024;
                d_tile_sums.reserve(nt);
                hipLaunchKernelGGL(k_chunk_scan_sums, dim3(nt), dim3(1024), 0, stream, g.K, 1, nullptr, d_counts.p, d_skip.p, d_extra.p, mm_p.cap, d_tile_sums.p,
                                   d_vout.p);
                hipLaunchKernelGGL(k_chunk_scan_apply, dim3(nt), dim3(1024), 0, stream, g.K, 1, nullptr, 1, nullptr, d_counts.p, d_skip.p, d_extra.p, d_tile_sums.p,
                                   d_seg.p, d_offsets.p, d_vout.p);
            }
            SD_HIP(hipMemcpyAsync(h_vout.p, d_vout.p, sizeof(VerdictOut), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipMemcpyAsync(&mm_s, d_mm_end.p + (g.K - 1), sizeof(mm_s), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            mm_s.inc -= n; // clock_recovery_mm.cpp:123-126
            if (mm_s.inc < 0)
                mm_s.inc = 0;
            if (h_vout.p->overflow)
                throw HipError("symbol scratch overflow");
            const long long tot = h_vout.p->total;
            const long long need_soft = is_bpsk ? tot : 2 * tot;
            if ((size_t)need_soft > soft_cap)
                throw HipError("soft output buffer too small");
            if (mm_p.q8)
                launch_compact8(symbuf.p, d_seg.p, d_offsets.p, g.K, mm_p.cap, is_bpsk ? 1 : 0, d_soft, (long long)soft_cap, stream);
            else
                launch_quantize(symbuf.p, d_seg.p, d_offsets.p, g.K, mm_p.cap, is_bpsk ? 1 : 0, d_soft, (long long)soft_cap, d_syms, (long long)syms_cap, stream);
            // history for the next call: last DEMOD_HIST de-rotated Costas outputs
            launch_tail_copy(A, n, DEMOD_HIST, cg, mm_rot, order, d_hist.p, stream);
            SD_HIP(hipMemcpyAsync(hist.data(), d_hist.p, DEMOD_HIST * sizeof(cf32), hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            stats.symbols_out += tot;
            last_symbols = tot;
            return need_soft;
        }

        int64_t process(const void *d_in, size_t n_in, int fmt, int8_t *d_soft, size_t soft_cap, float *d_syms, size_t syms_cap)
        {
            StatsScope _ss(*this);
            SD_HIP(hipSetDevice(cfg.device));
            // SDHIP_DEBUG: host wall-clock of every stage incl. its certificate round trips
            const bool tdbg = getenv("SDHIP_DEBUG") != nullptr;
            auto tnow = [] { return std::chrono::steady_clock::now(); };
            auto t_prev = tnow();
            auto tick = [&](const char *what) {
                if (!tdbg)
                    return;
                const auto t = tnow();
                fprintf(stderr, "[sdhip] demod %-10s %7.3f ms (host wall)\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
                t_prev = t;
            };
            stats.chunks = stats.chunks_fixed = stats.chunks_rotated = stats.chunks_inexact = stats.chunks_forced = 0;
            if (n_in == 0)
                return 0;
            long long n = (long long)n_in;
            const size_t need = (size_t)n + 2 * DEMOD_HIST + 64; // 64 + 64 samples of slack behind the data: the lanes prefetch up to 32 past their range
            bufA.reserve(need);
            bufB.reserve(need);
            cf32 *A = bufA.p + DEMOD_HIST, *B = bufB.p + DEMOD_HIST;
            stats.samples_in += n;

            // ---- stage 0: format conversion (+ iq_swap). cf32 without swap is already the stage format: the first stage reads
            // the caller's buffer in place (16-byte aligned pointers only: the lanes move float4 blocks).
            const bool in_place = fmt == SDHIP_FMT_CF32 && !cfg.iq_swap && !cfg.dc_block && cfg.freq_shift == 0 && !cfg.doppler && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
            const cf32 *SRC = A;
            if (in_place)
                SRC = reinterpret_cast<const cf32 *>(d_in);
            else
                launch_convert(d_in, fmt, cfg.iq_swap, n, A, stream);
            if (cfg.dc_block && cfg.exact)
            {
                SD_HIP(hipMemcpyAsync(d_dc.p, &dc_s, sizeof(dc_s), hipMemcpyHostToDevice, stream));
                launch_dcblock_seq(A, B, n, d_dc.p, stream);
                SD_HIP(hipMemcpyAsync(&dc_s, d_dc.p, sizeof(dc_s), hipMemcpyDeviceToHost, stream));
                std::swap(A, B);
                SRC = A; // the resampler reads the DC-blocked samples (found by the fuzz on the host twin: it read the stage's input)
            }
            else if (cfg.dc_block)
            {
                dc_block_chunked(A, B, n, dc_s);
                std::swap(A, B);
                SRC = A;
            }
            // ---- freq_shift (module_demod_base.cpp:122-123): the rotator, between the DC block and the resampler. The reference calls it once
            // per source buffer (the file source is built with the module's final d_buffer_size, module_demod_base.cpp:111; file_source.cpp:29)
            // and the kernel renormalises per call: the stream position carries across calls.
            if (cfg.freq_shift != 0)
            {
                const int src_buf = d_buffer_size;
                if (cfg.exact)
                {
                    SD_HIP(hipMemcpyAsync(d_rot_state.p, &rot_s, sizeof(rot_s), hipMemcpyHostToDevice, stream));
                    launch_rotator_seq(A, B, n, d_rot_state.p, rot_dre, rot_dim, src_buf, (int)(rot_abs % src_buf), stream);
                    SD_HIP(hipMemcpyAsync(&rot_s, d_rot_state.p, sizeof(rot_s), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                else
                    launch_rotator_par(A, B, n, rot_abs, rot_fix, rot_mag_eps, src_buf, stream);
                rot_abs += n;
                std::swap(A, B);
                SRC = A;
                tick("freq_shift");
            }
            // ---- Doppler correction (module_demod_base.cpp:125-171): behind the frequency shift, in front of the resampler
            if (cfg.doppler)
            {
                const int src_buf = d_buffer_size;
                const long long first = (long long)src_buf - dop_pos; // samples left in the buffer in progress
                const int nstart = n > first ? (int)((n - first + src_buf - 1) / src_buf) : 0; // source buffers that START inside this call
                if ((size_t)nstart > dop_queue.size())
                    throw HipError("doppler: " + std::to_string(nstart) + " source buffers start in this call but only " + std::to_string(dop_queue.size()) +
                                   " targets are queued (sdhip_demod_doppler_targets)");
                std::vector<float> t(dop_queue.begin(), dop_queue.begin() + nstart);
                const float alpha = cfg.doppler_alpha;
                if (cfg.exact)
                {
                    d_dop.reserve(1);
                    d_dop_t.reserve(std::max<size_t>(1, t.size()));
                    SD_HIP(hipMemcpyAsync(d_dop.p, &dop_s, sizeof(dop_s), hipMemcpyHostToDevice, stream));
                    if (!t.empty())
                        SD_HIP(hipMemcpyAsync(d_dop_t.p, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, stream));
                    launch_doppler_seq(A, B, n, d_dop.p, alpha, dop_target, d_dop_t.p, src_buf, dop_pos, stream);
                    SD_HIP(hipMemcpyAsync(&dop_s, d_dop.p, sizeof(dop_s), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                else
                {
                    // start states of the buffers touched, buffer to buffer in closed form (double)
                    const double a = (double)alpha, lb = log1p(-a);
                    std::vector<DopStart> st;
                    double ph = dop_ph, f = dop_f, targ = (double)dop_target;
                    long long left = n;
                    long long seg = std::min<long long>(left, first);
                    size_t tk = 0;
                    for (;;)
                    {
                        if (seg == 0 && left > 0)
                        { // the buffer in progress is used up: the next sample opens a new one
                            targ = (double)t[tk++];
                            seg = std::min<long long>(left, src_buf);
                        }
                        st.push_back(DopStart{ph, f, targ});
                        const double bk = exp(lb * (double)seg);
                        ph = ph + (double)seg * targ + (f - targ) * (1.0 - bk) / a;
                        ph -= 2.0 * design::PI * rint(ph / (2.0 * design::PI));
                        f = targ + (f - targ) * bk;
                        left -= seg;
                        if (left <= 0)
                            break;
                        seg = 0;
                    }
                    // the kernel indexes buffers by (i + pos0) / buf_len: an exhausted buffer in progress (first == 0) occupies index 0 with no samples
                    if (first == 0)
                        st.insert(st.begin(), DopStart{dop_ph, dop_f, (double)dop_target});
                    d_dop_starts.reserve(st.size());
                    SD_HIP(hipMemcpyAsync(d_dop_starts.p, st.data(), st.size() * sizeof(DopStart), hipMemcpyHostToDevice, stream));
                    launch_doppler_par(A, B, n, d_dop_starts.p, a, src_buf, dop_pos, stream);
                    SD_HIP(hipStreamSynchronize(stream)); // (st is a local)
                    dop_ph = ph;
                    dop_f = f;
                }
                for (int k = 0; k < nstart; k++)
                    dop_queue.pop_front();
                if (nstart)
                    dop_target = t.back();
                dop_pos = nstart ? (int)(n - first - (long long)(nstart - 1) * src_buf) : (int)(dop_pos + n);
                std::swap(A, B);
                SRC = A;
                tick("doppler");
            }
            // ---- SmartResampler: power-of-two pre-decimator stages (if the ratio has them), then the rational resampler
            bool in_place_r = in_place;
            if (resample && !pd_stages.empty())
            {
                const cf32 *cur = SRC;
                long long ncur = n;
                for (auto &stp : pd_stages)
                {
                    DecimStage &ds = *stp;
                    cf32 *dst = (cur == A) ? B : A;
                    const long long nout = ncur > ds.inc ? (ncur - ds.inc + ds.decim - 1) / ds.decim : 0;
                    launch_decim_fir(cur, ds.d_hist.p, ncur, ds.d_taps.p, ds.ntaps, ds.decim, ds.inc, dst, nout, stream);
                    hipLaunchKernelGGL(k_hist_slide, dim3(1), dim3(1024), 0, stream, ds.d_hist.p, ds.ntaps, cur, ncur);
                    ds.inc = (int)(ds.inc + nout * ds.decim - ncur); // decimating_fir.cpp:84
                    cur = dst;
                    ncur = nout;
                }
                if (cur != A)
                    std::swap(A, B);
                SRC = A;
                in_place_r = false;
                n = ncur;
                if (n == 0)
                    return 0;
            }
            if (resample && rational)
            {
                const bool in_place = in_place_r; // the rational stage reads the caller's buffer only when nothing ran in front of it
                if (in_place)
                    SD_HIP(hipMemcpyAsync(d_hist_in.p, hist_in.data(), DEMOD_HIST * sizeof(cf32), hipMemcpyHostToDevice, stream));
                else
                    put_hist(A, hist_in);
                // outputs m with inc0 + (ctr0 + m*decim)/interp < n
                const long long lim = (n - r_inc) * (long long)r_interp - r_ctr;
                const long long nout = lim > 0 ? (lim + r_decim - 1) / r_decim : 0;
                ResampParams rp{(int)r_interp, (int)r_decim, r_ntaps, d_rbank.p};
                launch_resample(SRC, in_place ? d_hist_in.p : A - DEMOD_HIST, n, rp, r_ctr, r_inc, B, nout, stream);
                if (in_place && n >= DEMOD_HIST)
                {
                    SD_HIP(hipMemcpyAsync(hist_in.data(), SRC + n - DEMOD_HIST, DEMOD_HIST * sizeof(cf32), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                else if (in_place)
                { // short call: slide the host history
                    std::vector<cf32> t(n);
                    SD_HIP(hipMemcpyAsync(t.data(), SRC, (size_t)n * sizeof(cf32), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                    hist_in.erase(hist_in.begin(), hist_in.begin() + n);
                    hist_in.insert(hist_in.end(), t.begin(), t.end());
                }
                else
                    get_hist(A, n, hist_in);
                const long long ph_end = r_ctr + nout * (long long)r_decim;
                r_inc = (int)(r_inc + ph_end / r_interp - n);
                r_ctr = (int)(ph_end % r_interp);
                std::swap(A, B);
                n = nout;
                if (n == 0)
                    return 0;
            }
            tick("resample");

            ChunkGeom cg;
            if (fuse_afc)
            {
                const cf32 *AIN = (in_place && !resample) ? SRC : A; // the lanes never load outside [chunk begin - warm-up, chunk end)
                afc_chunked(AIN, B, n, cg);
                std::swap(A, B);
                tick("agc+fir+costas");
            }
            // ---- AGC (speculative)
            if (!fuse_afc)
            {
                const cf32 *AIN = (in_place && !resample) ? SRC : A; // k_chunks never loads outside [chunk begin - W, chunk end)
                agc_stage(AIN, B, n);
                std::swap(A, B);
            }
            tick("agc");
            // ---- RRC FIR (parallel, exact) -- unless it rode on the AGC lanes
            if (!fuse_agc_fir)
            {
                put_hist(A, hist_agc);
                launch_fir(A, B, n, d_rrc.p, rrc_ntaps, stream);
                get_hist(A, n, hist_agc);
                std::swap(A, B);
            }

            tick("fir");
            // ---- has_carrier (module_psk_demod.cpp:93-113): carrier PLL, then the DC block that takes the carrier line out
            if (cfg.has_carrier)
            {
                carrier_pll_chunked(A, B, n);
                std::swap(A, B);
                if (cfg.exact)
                {
                    SD_HIP(hipMemcpyAsync(d_dc.p, &dcc_s, sizeof(dcc_s), hipMemcpyHostToDevice, stream));
                    launch_dcblock_seq(A, B, n, d_dc.p, stream);
                    SD_HIP(hipMemcpyAsync(&dcc_s, d_dc.p, sizeof(dcc_s), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                else
                    dc_block_chunked(A, B, n, dcc_s);
                std::swap(A, B);
                tick("carrier");
            }
            // ---- Costas (speculative, symmetry-corrected)
            if (!fuse_afc && !nd.skip_costas)
            {
                costas_stage(A, B, n, cg, final_sps, (double)final_samplerate);
                std::swap(A, B);
            }
            // ---- post_costas_dc (module_psk_demod.cpp:127-134): the DC block sees ONE coherent stream, so the per-chunk frames of the
            // Costas stage are turned back first (exact quarter / half turns); the clock recovery then reads un-rotated samples
            const int *mm_rot = nd.skip_costas ? nullptr : d_rot.p; // no Costas stage: no per-chunk frames to undo
            if (nd.skip_costas)
                cg = make_geom(n, 1 << 30, 0);
            if (cfg.post_costas_dc)
            {
                launch_derotate(A, n, cg, d_rot.p, order, stream);
                if (cfg.exact)
                {
                    SD_HIP(hipMemcpyAsync(d_dc.p, &dc2_s, sizeof(dc2_s), hipMemcpyHostToDevice, stream));
                    launch_dcblock_seq(A, B, n, d_dc.p, stream);
                    SD_HIP(hipMemcpyAsync(&dc2_s, d_dc.p, sizeof(dc2_s), hipMemcpyDeviceToHost, stream));
                    SD_HIP(hipStreamSynchronize(stream));
                }
                else
                    dc_block_chunked(A, B, n, dc2_s);
                std::swap(A, B);
                mm_rot = nullptr;
            }
            tick("costas");
            // ---- M&M + quantiser
            const int64_t nsoft = mm_stage(A, n, cg, mm_rot, d_soft, soft_cap, d_syms, syms_cap, hist_cos);
            tick("mm+quant");
            started = true;
            return nsoft;
        }

        // ---- ndsp: PSKDemodHierBlock's chain (dsp/hier/psk_demod.h:60-66: agc <- rrc, rec <- agc, pll <- rec). d_in: n complex floats,
        // d_out: the symbols (complex floats). Same lane-per-chunk stages as process(); the Costas loop runs over the clock recovery's
        // SYMBOLS and its per-chunk frames are turned back on its output (exact quarter / half turns).
        int64_t process_ndsp(const float *d_in, size_t n_in, float *d_out, size_t out_cap)
        {
            SD_HIP(hipSetDevice(cfg.device));
            stats.chunks = stats.chunks_fixed = stats.chunks_rotated = stats.chunks_inexact = stats.chunks_forced = 0;
            if (n_in == 0)
                return 0;
            long long n = (long long)n_in;
            const size_t need = (size_t)n + 2 * DEMOD_HIST + 64;
            bufA.reserve(need);
            bufB.reserve(need);
            cf32 *A = bufA.p + DEMOD_HIST, *B = bufB.p + DEMOD_HIST;
            stats.samples_in += n;
            if (nd.only == SDHIP_NDSP_AGC || nd.only == SDHIP_NDSP_AGC_FAST)
            { // AGCBlock<complex_t>::process (dsp/agc/agc.cpp:22-39) / AGCFastBlock<complex_t>::process (dsp/agc/agc_fast.cpp:22-58) on its own
                if ((size_t)n > out_cap)
                    throw HipError("output buffer too small");
                SD_HIP(hipMemcpyAsync(A, d_in, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                agc_stage(A, B, n);
                SD_HIP(hipMemcpyAsync(d_out, B, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                SD_HIP(hipStreamSynchronize(stream));
                started = true;
                return n;
            }
            if (nd.only == SDHIP_NDSP_COSTAS_FAST)
            { // CostasFastBlock::process (dsp/pll/costas_fast.cpp:93-106): costas_fast_stage
                if ((size_t)n > out_cap)
                    throw HipError("output buffer too small");
                // (the lanes read [chunk start - warm-up, chunk end) of the call and nothing in front of it: the caller's buffers serve as they are when they
                // sit on 16 bytes -- the lanes move 64-byte blocks --, which saves two passes over the call)
                if (((uintptr_t)d_in % 16) == 0 && ((uintptr_t)d_out % 16) == 0 && (const void *)d_in != (const void *)d_out)
                    costas_fast_stage(reinterpret_cast<const cf32 *>(d_in), reinterpret_cast<cf32 *>(d_out), n);
                else
                {
                    SD_HIP(hipMemcpyAsync(A, d_in, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                    costas_fast_stage(A, B, n);
                    SD_HIP(hipMemcpyAsync(d_out, B, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                }
                SD_HIP(hipStreamSynchronize(stream));
                stats.freq_hz = (float)(((double)cf_s.freq / (2.0 * design::PI)) * nd.samplerate); // the block's "freq" statistic is rad / sample (costas_fast.h:80)
                started = true;
                return n;
            }
            if (nd.only == SDHIP_NDSP_MM || nd.only == SDHIP_NDSP_GARDNER || nd.only == SDHIP_NDSP_MM_FAST)
            { // MMClockRecoveryBlock<complex_t>::work (dsp/clock_recovery/clock_recovery_mm.cpp:66-183) on its own; GardnerClockRecoveryBlock<complex_t>::work
              // (dsp/clock_recovery/clock_recovery_gardner.cpp:60-170) on the same lanes with its own iteration (mm_p.loop)
                SD_HIP(hipMemcpyAsync(A, d_in, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                if (nd.only == SDHIP_NDSP_MM_FAST)
                { // lane per (chunk, cadence) with the bit-exact hand-off; -1: this call runs as the one sequential lane below
                    const long long r = mmfast_stage(A, n, d_out, out_cap);
                    if (r >= 0)
                    {
                        started = true;
                        return r;
                    }
                }
                const double omin1 = (double)mm_p.omega_mid - (double)mm_p.omega_limit;
                const size_t symcap1 = (size_t)((double)n / std::max(0.5, omin1 - 0.01)) + 64;
                if (symcap1 > out_cap + 64 && (size_t)((double)n / omin1) + 8 > out_cap)
                    throw HipError("symbol output buffer too small");
                symtmp.reserve(symcap1 + 2 * DEMOD_HIST + 64);
                d_soft_tmp.reserve(2 * symcap1);
                cf32 *S1 = symtmp.p + DEMOD_HIST;
                const ChunkGeom one1 = make_geom(n, 1 << 30, 0);
                mm_stage(A, n, one1, nullptr, d_soft_tmp.p, 2 * symcap1, reinterpret_cast<float *>(S1), symcap1, hist_cos);
                const long long ns1 = last_symbols;
                if ((size_t)ns1 > out_cap)
                    throw HipError("symbol output buffer too small");
                if (ns1)
                    SD_HIP(hipMemcpyAsync(d_out, S1, (size_t)ns1 * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                SD_HIP(hipStreamSynchronize(stream));
                started = true;
                return ns1;
            }
            if (nd.only == SDHIP_NDSP_COSTAS)
            { // CostasBlock::process (dsp/pll/costas.cpp:12-61) on its own: the loop over whatever rate its input has
                if ((size_t)n > out_cap)
                    throw HipError("output buffer too small");
                symtmp.reserve((size_t)n + 2 * DEMOD_HIST + 64);
                cf32 *S1 = symtmp.p + DEMOD_HIST;
                SD_HIP(hipMemcpyAsync(S1, d_in, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                ChunkGeom cg1;
                costas_stage(S1, B, n, cg1, 1.0, nd.samplerate);
                launch_derotate(B, n, cg1, d_rot.p, order, stream);
                SD_HIP(hipMemcpyAsync(d_out, B, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                SD_HIP(hipStreamSynchronize(stream));
                started = true;
                return n;
            }
            // ---- RRC FIR (FIRBlock::process, dsp/filter/fir.cpp:62-133): same dot products as the legacy block's (the aligned kernel call
            // only puts zero taps in front), but the block holds ntaps samples back: output i is the window starting at input i + 1, and
            // the first call returns ntaps samples fewer. In stream terms: the legacy filter's output without its first ntaps samples.
            SD_HIP(hipMemcpyAsync(A, d_in, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
            put_hist(A, hist_in);
            launch_fir(A, B, n, d_rrc.p, rrc_ntaps, stream);
            get_hist(A, n, hist_in);
            std::swap(A, B);
            if (fir_drop > 0)
            {
                const long long d = std::min<long long>(fir_drop, n);
                fir_drop -= d;
                n -= d;
                if (n == 0)
                    return 0;
                SD_HIP(hipMemcpyAsync(B, A + d, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                std::swap(A, B);
            }
            if (nd.only == SDHIP_NDSP_RRC_FIR)
            { // RRC_Block<FIRBlock<complex_t>> on its own
                if ((size_t)n > out_cap)
                    throw HipError("output buffer too small");
                SD_HIP(hipMemcpyAsync(d_out, A, (size_t)n * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
                SD_HIP(hipStreamSynchronize(stream));
                started = true;
                return n;
            }
            // ---- AGC (AGCBlock::process, dsp/agc/agc.cpp:22-39), reference 0.6 from the hier block's constructor
            agc_stage(A, B, n);
            std::swap(A, B);
            // ---- M&M (MMClockRecoveryBlock::work, dsp/clock_recovery/clock_recovery_mm.cpp:66-183): symbols as floats into symtmp
            const double omin = (double)mm_p.omega_mid - (double)mm_p.omega_limit;
            const size_t symcap = (size_t)((double)n / std::max(0.5, omin - 0.01)) + 64;
            symtmp.reserve(symcap + 2 * DEMOD_HIST + 64);
            d_soft_tmp.reserve(2 * symcap);
            cf32 *S = symtmp.p + DEMOD_HIST;
            const ChunkGeom one = make_geom(n, 1 << 30, 0);
            mm_stage(A, n, one, nullptr, d_soft_tmp.p, 2 * symcap, reinterpret_cast<float *>(S), symcap, hist_cos);
            const long long nsym = last_symbols;
            if (nsym == 0)
            {
                started = true;
                return 0;
            }
            if ((size_t)nsym > out_cap)
                throw HipError("symbol output buffer too small");
            // ---- Costas loop over the symbols (CostasBlock::process, dsp/pll/costas.cpp:12-61), frames turned back
            cf32 *O = B; // the free stage buffer (nsym <= n)
            ChunkGeom cg;
            costas_stage(S, O, nsym, cg, 1.0, nd.symbolrate);
            launch_derotate(O, nsym, cg, d_rot.p, order, stream);
            SD_HIP(hipMemcpyAsync(d_out, O, (size_t)nsym * sizeof(cf32), hipMemcpyDeviceToDevice, stream));
            SD_HIP(hipStreamSynchronize(stream));
            started = true;
            return nsym;
        }

        // ---- host path
        // one staged batch: shipped into a device slot by the pipe's first thread (its own copy stream), processed by the second (the stages, D2H of the
        // soft symbols) while the next batch is on its way
        DevBuf<uint8_t> d_in_slot[2];
        hipStream_t copy_stream = nullptr;
        void ship_in(const uint8_t *pinned, size_t bytes, int /*fmt*/, int slot)
        {
            SD_HIP(hipSetDevice(cfg.device));
            if (!copy_stream)
                SD_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
            d_in_slot[slot].reserve(bytes);
            SD_HIP(hipMemcpyAsync(d_in_slot[slot].p, pinned, bytes, hipMemcpyHostToDevice, copy_stream));
            SD_HIP(hipStreamSynchronize(copy_stream));
        }
        void ship(size_t bytes, int fmt, int slot)
        {
            const size_t ns = bytes / fmt_bytes(fmt);
            if (ns == 0)
                return;
            SD_HIP(hipSetDevice(cfg.device));
            const size_t cap = 2 * ns + 64;
            d_soft_tmp.reserve(cap);
            const int64_t n = process(d_in_slot[slot].p, ns, fmt, d_soft_tmp.p, cap, nullptr, 0);
            h_out.reserve((size_t)n + 1);
            SD_HIP(hipMemcpyAsync(h_out.p, d_soft_tmp.p, (size_t)n, hipMemcpyDeviceToHost, stream));
            SD_HIP(hipStreamSynchronize(stream));
            OutChunk c{std::unique_ptr<int8_t[]>(new int8_t[(size_t)n + 1]), (size_t)n};
            CopyPool::get().copy(c.p.get(), h_out.p, (size_t)n);
            std::lock_guard<std::mutex> lk(out_mu);
            out_queue.push_back(std::move(c));
        }
        int push_host(const void *iq, size_t nsamples, int fmt)
        {
            if (!pipe)
                pipe.reset(new HostPipe([this](const uint8_t *p, size_t b, int f, int s) { ship_in(p, b, f, s); }, [this](size_t b, int f, int s) { ship(b, f, s); }));
            const size_t bps = (size_t)fmt_bytes(fmt);
            pipe->push(iq, nsamples * bps, fmt, HOST_BATCH * bps);
            return 0;
        }
        int flush_host()
        {
            if (pipe)
                pipe->flush();
            return 0;
        }
        int64_t pull(int8_t *soft, size_t cap)
        {
            size_t done = 0;
            while (done < cap)
            {
                const int8_t *src;
                size_t take;
                {
                    std::lock_guard<std::mutex> lk(out_mu);
                    if (out_queue.empty())
                        break;
                    const OutChunk &c = out_queue.front();
                    take = std::min(c.n - out_read, cap - done);
                    src = c.p.get() + out_read; // the front chunk stays where it is while this (single) consumer copies out of it
                }
                CopyPool::get().copy(soft + done, src, take);
                done += take;
                std::lock_guard<std::mutex> lk(out_mu);
                out_read += take;
                if (out_read == out_queue.front().n)
                {
                    out_queue.pop_front();
                    out_read = 0;
                }
            }
            return (int64_t)done;
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

namespace sdhip
{
    // internal (csrc/dvbs2_engine.hip): the clock recovery's hand-off windows of one handle, in samples -- `tight` = what the adaptive warm-up is measured against,
    // `tol` = beyond it a boundary is re-run from the exact state
    void demod_set_mm_windows(void *h, double tight, double tol)
    {
        DemodEngine *e = (DemodEngine *)h;
        e->mm_windows_tight = tight;
        e->mm_windows_tol = tol;
    }
} // namespace sdhip

extern "C"
{
    void sdhip_demod_cfg_default(sdhip_demod_cfg *c)
    {
        memset(c, 0, sizeof(*c));
        c->constellation = SDHIP_QPSK;
        c->rrc_taps = 31;
        c->agc_rate = 1e-2f;
        c->min_sps = 1.1f;
        c->max_sps = 4.0f;
        c->clock_gain_omega = (float)(pow(8.7e-3, 2) / 4.0); // module_psk_demod.h:36-39
        c->clock_mu = 0.5f;
        c->clock_gain_mu = (float)8.7e-3;
        c->clock_omega_relative_limit = 0.005f;
        c->carrier_pll_max_offset = 3.14f; // module_psk_demod.cpp:104
        c->doppler_alpha = 0.01f;          // module_demod_base.h:61
    }
    void *sdhip_demod_create(const sdhip_demod_cfg *cfg)
    {
        SD_GUARD_BEGIN
        return new DemodEngine(*cfg);
        SD_GUARD_END(nullptr)
    }
    void sdhip_demod_destroy(void *h) { delete (DemodEngine *)h; }
    int sdhip_demod_push(void *h, const void *iq, size_t nsamples, int fmt)
    {
        SD_GUARD_BEGIN
        return ((DemodEngine *)h)->push_host(iq, nsamples, fmt);
        SD_GUARD_END(-1)
    }
    int sdhip_demod_flush(void *h)
    {
        SD_GUARD_BEGIN
        return ((DemodEngine *)h)->flush_host();
        SD_GUARD_END(-1)
    }
    int64_t sdhip_demod_pull(void *h, int8_t *soft, size_t cap)
    {
        SD_GUARD_BEGIN
        return ((DemodEngine *)h)->pull(soft, cap);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_demod_process_dev(void *h, const void *d_iq, size_t nsamples, int fmt, int8_t *d_soft, size_t soft_cap, float *d_syms, size_t syms_cap, int final)
    {
        (void)final; // the demodulator has no block granularity: every call consumes all its samples
        SD_GUARD_BEGIN
        return ((DemodEngine *)h)->process(d_iq, nsamples, fmt, d_soft, soft_cap, d_syms, syms_cap);
        SD_GUARD_END(-1)
    }
    // ---- ndsp PSK demodulator (PSKDemodHierBlock) on the same engine
    void sdhip_ndsp_psk_cfg_default(sdhip_ndsp_psk_cfg *c)
    {
        memset(c, 0, sizeof(*c));
        c->constellation = SDHIP_BPSK; // psk_demod.h:34-37
        c->samplerate = 6e6;
        c->symbolrate = 2e6;
        c->rrc_gain = 1; // rrc.h:17-21
        c->rrc_alpha = 0.35;
        c->rrc_ntaps = 31;
        c->agc_rate = 1e-4f; // agc.h:14-17, reference from psk_demod.cpp:13
        c->agc_reference = 0.6f;
        c->agc_gain = 1.0f;
        c->agc_max_gain = 65536.0f;
        c->rec_omega = 0.0f;
        c->rec_omegaGain = (float)(pow(8.7e-3, 2) / 4.0); // clock_recovery_mm.h:17-23
        c->rec_mu = 0.5f;
        c->rec_muGain = (float)8.7e-3;
        c->rec_omegaLimit = (float)0.005;
        c->rec_nfilt = 128;
        c->rec_ntaps = 8;
        c->pll_loop_bw = (float)0.004; // costas.h:14-16
        c->pll_freq_limit = 1.0f;
    }
    static void *ndsp_create(const sdhip_ndsp_psk_cfg *c, int kind);
    void *sdhip_ndsp_psk_demod_create(const sdhip_ndsp_psk_cfg *c) { return ndsp_create(c, SDHIP_NDSP_HIER); }
    static void *ndsp_create(const sdhip_ndsp_psk_cfg *c, int kind)
    {
        SD_GUARD_BEGIN
        // the hier block: bpsk / qpsk (set_cfg returns RES_ERR for anything else, psk_demod.h:205-214); CostasBlock on its own also has order 8 (costas.h:78)
        if (c->constellation != SDHIP_BPSK && c->constellation != SDHIP_QPSK && !((kind == SDHIP_NDSP_COSTAS || kind == SDHIP_NDSP_COSTAS_FAST) && c->constellation == SDHIP_8PSK))
            throw HipError("ndsp psk_demod: constellation must be bpsk or qpsk");
        if (c->rec_nfilt != 128 || c->rec_ntaps != 8)
            throw HipError("ndsp psk_demod: the HIP path carries the 128 x 8 interpolator bank only");
        if (!(c->samplerate > 0) || !(c->symbolrate > 0))
            throw HipError("ndsp psk_demod: samplerate and symbolrate must be set");
        // the stage buffers hold one symbol per input sample at most: an omega below one sample per symbol (the advanced "rec_omega" key; 0 = samplerate /
        // symbolrate) would have the clock recovery write more symbols than samples came in
        if (c->rec_omega != 0.0f && !(c->rec_omega * (1.0f - fabsf(c->rec_omegaLimit)) >= 1.0f))
            throw HipError("ndsp psk_demod: rec_omega (with rec_omegaLimit) must stay at or above one sample per symbol");
        if (c->rec_omega == 0.0f && !(c->samplerate / c->symbolrate * (1.0 - fabs((double)c->rec_omegaLimit)) >= 1.0))
            throw HipError("ndsp psk_demod: samplerate / symbolrate (with rec_omegaLimit) must stay at or above one sample per symbol");
        sdhip_demod_cfg d;
        sdhip_demod_cfg_default(&d);
        d.device = c->device;
        d.constellation = c->constellation;
        d.samplerate = (decltype(d.samplerate))c->samplerate;
        d.symbolrate = (decltype(d.symbolrate))c->symbolrate;
        d.min_sps = 1.0f; // never resample: the hier block has no resampler
        d.max_sps = 3.0e38f;
        d.rrc_alpha = (float)c->rrc_alpha;
        d.rrc_taps = c->rrc_ntaps;
        d.agc_rate = c->agc_rate;
        d.pll_bw = c->pll_loop_bw;
        d.clock_gain_omega = c->rec_omegaGain;
        d.clock_mu = c->rec_mu;
        d.clock_gain_mu = c->rec_muGain;
        d.clock_omega_relative_limit = c->rec_omegaLimit;
        d.exact = c->exact;
        d.chunk_len = c->chunk_len;
        d.warmup = c->warmup;
        NdspExt e;
        e.on = true;
        e.samplerate = c->samplerate;
        e.symbolrate = c->symbolrate;
        e.rrc_gain = c->rrc_gain;
        e.rrc_alpha = c->rrc_alpha;
        e.agc_reference = c->agc_reference;
        e.agc_gain = c->agc_gain;
        e.agc_max_gain = c->agc_max_gain;
        e.rec_omega = c->rec_omega;
        e.pll_freq_limit = c->pll_freq_limit;
        e.only = kind;
        DemodEngine *eng = new DemodEngine(d, &e);
        if (kind != SDHIP_NDSP_HIER && kind != SDHIP_NDSP_RRC_FIR)
            eng->fir_drop = 0;
        return eng;
        SD_GUARD_END(nullptr)
    }
    void sdhip_ndsp_psk_demod_destroy(void *h) { delete (DemodEngine *)h; }
    void *sdhip_ndsp_block_create(int kind, const sdhip_ndsp_psk_cfg *c)
    {
        if (kind < SDHIP_NDSP_HIER || kind > SDHIP_NDSP_MM_FAST)
        {
            sdhip::set_error("ndsp block: unknown kind");
            return nullptr;
        }
        return ndsp_create(c, kind);
    }
    // DVB-S2 demodulator front: a psk_demod handle without the Costas loop (use sdhip_demod_process_dev with d_syms for the symbols)
    void *sdhip_dvbs2_front_create(const sdhip_demod_cfg *cfg)
    {
        SD_GUARD_BEGIN
        if (cfg->has_carrier || cfg->post_costas_dc)
            throw HipError("dvbs2 front: has_carrier / post_costas_dc belong to psk_demod's carrier loop");
        NdspExt e;
        e.skip_costas = true;
        return new DemodEngine(*cfg, &e);
        SD_GUARD_END(nullptr)
    }
    int64_t sdhip_ndsp_psk_demod_work_dev(void *h, const float *d_in, size_t nsamples, float *d_out, size_t out_cap)
    {
        SD_GUARD_BEGIN
        DemodEngine *e = (DemodEngine *)h;
        if (!e->nd.on)
            throw HipError("not an ndsp demodulator handle");
        return e->process_ndsp(d_in, nsamples, d_out, out_cap);
        SD_GUARD_END(-1)
    }
    int64_t sdhip_ndsp_psk_demod_work(void *h, const float *in, size_t nsamples, float *out, size_t out_cap)
    {
        SD_GUARD_BEGIN
        DemodEngine *e = (DemodEngine *)h;
        if (!e->nd.on)
            throw HipError("not an ndsp demodulator handle");
        SD_HIP(hipSetDevice(e->cfg.device));
        DevBuf<float> din, dout;
        din.reserve(2 * nsamples + 32);
        dout.reserve(2 * out_cap + 32);
        SD_HIP(hipMemcpy(din.p, in, 2 * nsamples * sizeof(float), hipMemcpyHostToDevice));
        const int64_t r = e->process_ndsp(din.p, nsamples, dout.p, out_cap);
        if (r > 0)
            SD_HIP(hipMemcpy(out, dout.p, 2 * (size_t)r * sizeof(float), hipMemcpyDeviceToHost));
        return r;
        SD_GUARD_END(-1)
    }
    int sdhip_ndsp_psk_demod_get_stats(void *h, sdhip_demod_stats *st) { return sdhip_demod_get_stats(h, st); }
    int sdhip_demod_doppler_targets(void *h, const float *targets, size_t n)
    {
        SD_GUARD_BEGIN
        DemodEngine *e = (DemodEngine *)h;
        if (!e->cfg.doppler)
            throw HipError("doppler targets handed to a handle without cfg.doppler");
        for (size_t i = 0; i < n; i++)
            e->dop_queue.push_back(targets[i]);
        return 0;
        SD_GUARD_END(-1)
    }
    int sdhip_demod_get_stats(void *h, sdhip_demod_stats *st)
    {
        DemodEngine *e = (DemodEngine *)h;
        std::lock_guard<std::mutex> lk(e->stats_mu);
        *st = e->stats_busy > 0 ? e->stats_pub : e->stats;
        return 0;
    }
    int sdhip_demod_set_tap(void *h, int mode)
    {
        SD_GUARD_BEGIN
        auto *e = (DemodEngine *)h;
        if (mode != 0 && mode != 1)
            throw HipError("unknown tap mode");
        if (mode && e->cfg.exact)
            throw HipError("the arm tap exists for the chunk-parallel mode only (the exact mode IS the reference's trajectory)");
        e->tap_mode = mode;
        return 0;
        SD_GUARD_END(-1)
    }

    // the coefficient tables the modules are built with (host only: no device involved)
    int64_t sdhip_design(int kind, const double *params, float *out, size_t cap, int *dims)
    {
        SD_GUARD_BEGIN
        std::vector<float> t;
        int d0 = 0, d1 = 0;
        if (kind == 0)
        {
            t = design::rrc(params[0], params[1], params[2], params[3], (int)params[4]);
            d0 = (int)t.size();
        }
        else if (kind == 1)
        {
            d0 = (int)params[0];
            d1 = design::mm_bank((int)params[0], (int)params[1], t);
        }
        else if (kind == 2)
        {
            unsigned ip = (unsigned)params[0], dc = (unsigned)params[1];
            const int nt = design::resampler_bank(ip, dc, t);
            d0 = (int)ip;
            d1 = (int)dc;
            if (dims)
                dims[2] = nt;
        }
        else
            throw HipError("unknown table kind");
        if (t.size() > cap)
            throw HipError("output too small");
        memcpy(out, t.data(), t.size() * sizeof(float));
        if (dims)
        {
            dims[0] = d0;
            dims[1] = d1;
        }
        return (int64_t)t.size();
        SD_GUARD_END(-1)
    }

    // single blocks, exact sequential semantics (one lane): arithmetic parity of each kernel body
    int64_t sdhip_op_block(int device, int kind, const float *params, const float *d_in, size_t n, float *d_out, size_t out_cap)
    {
        SD_GUARD_BEGIN
        SD_HIP(hipSetDevice(device));
        const long long nn = (long long)n;
        DevBuf<cf32> in;
        in.reserve(n + 2 * DEMOD_HIST + 64);
        SD_HIP(hipMemset(in.p, 0, DEMOD_HIST * sizeof(cf32)));
        cf32 *X = in.p + DEMOD_HIST;
        SD_HIP(hipMemcpy(X, d_in, n * sizeof(cf32), hipMemcpyDeviceToDevice));
        cf32 *Y = (cf32 *)d_out;
        const ChunkGeom g = make_geom(nn, 1 << 30, 0);
        int64_t nout = nn;
        if (kind == 0)
        {
            if (out_cap < n)
                throw HipError("output too small");
            AgcParams p{params[0], params[1], params[3], params[2]};
            AgcState s0{params[2]};
            DevBuf<AgcState> st;
            st.reserve(3);
            SD_HIP(hipMemcpy(st.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            launch_agc(X, Y, g, p, st.p, st.p + 1, st.p + 2, nullptr, 0, nullptr);
        }
        else if (kind == 1)
        {
            std::vector<float> t = design::rrc(1, params[0], params[1], params[2], (int)params[3]);
            std::vector<float> r(t.rbegin(), t.rend());
            DevBuf<float> dt;
            dt.reserve(r.size());
            SD_HIP(hipMemcpy(dt.p, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
            launch_fir(X, Y, nn, dt.p, (int)r.size(), nullptr);
        }
        else if (kind == 2 || kind == 9)
        { // 9: the ndsp CostasBlock (dsp/pll/costas.cpp:12-61) -- the legacy loop with dsp::branched_clip on the error
            CostasParams p{};
            p.clip_branched = kind == 9 ? 1 : 0;
            design::costas_gains(params[0], p.alpha, p.beta);
            p.order = (int)params[1];
            p.fmin = -params[2];
            p.fmax = params[2];
            CostasState s0{0, 0};
            DevBuf<CostasState> st;
            st.reserve(3);
            SD_HIP(hipMemcpy(st.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            launch_costas(X, Y, g, p, st.p, st.p + 1, st.p + 2, nullptr, 0, nullptr);
        }
        else if (kind == 3)
        {
            std::vector<float> mmb;
            design::mm_bank(128, 8, mmb);
            DevBuf<float> db;
            db.reserve(mmb.size());
            SD_HIP(hipMemcpy(db.p, mmb.data(), mmb.size() * sizeof(float), hipMemcpyHostToDevice));
            MmParams p{};
            p.omega_gain = params[1];
            p.mu_gain = params[3];
            p.omega_mid = params[0];
            p.omega_limit = params[4] * params[0];
            p.init_mu = params[2];
            p.bank = db.p;
            p.cap = (int)out_cap;
            p.cg = g;
            p.rot = nullptr;
            p.order = 4;
            MmState s0{};
            s0.mu = params[2];
            s0.omega = params[0];
            DevBuf<MmState> st;
            st.reserve(3);
            DevBuf<int> cnt;
            cnt.reserve(2);
            SD_HIP(hipMemcpy(st.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            DevBuf<MmCert> cc;
            cc.reserve(2);
            launch_mm(X, Y, cnt.p, g, p, st.p, st.p + 1, st.p + 2, cc.p, cc.p + 1, nullptr, 0, nullptr);
            int c = 0;
            SD_HIP(hipMemcpy(&c, cnt.p, sizeof(int), hipMemcpyDeviceToHost));
            nout = c;
        }
        else if (kind == 4)
        {
            unsigned ip = (unsigned)params[0], dc = (unsigned)params[1];
            std::vector<float> bank;
            const int nt = design::resampler_bank(ip, dc, bank);
            DevBuf<float> db;
            db.reserve(bank.size());
            SD_HIP(hipMemcpy(db.p, bank.data(), bank.size() * sizeof(float), hipMemcpyHostToDevice));
            const long long lim = nn * (long long)ip;
            nout = (lim + dc - 1) / dc;
            if ((size_t)nout > out_cap)
                throw HipError("output too small");
            ResampParams rp{(int)ip, (int)dc, nt, db.p};
            launch_resample(X, X - DEMOD_HIST, nn, rp, 0, 0, Y, nout, nullptr);
        }
        else if (kind == 5)
        { // CorrectIQBlock (DC block), correct_iq.cpp:18-35
            if (out_cap < n)
                throw HipError("output too small");
            DcState s0{0, 0};
            DevBuf<DcState> st;
            st.reserve(1);
            SD_HIP(hipMemcpy(st.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            launch_dcblock_seq(X, Y, nn, st.p, nullptr);
        }
        else if (kind == 7 || kind == 10)
        { // GardnerClockRecoveryBlock(omega, omega_gain, mu, mu_gain, omega_limit), clock_recovery_gardner.cpp:10-24; 10: the ndsp block's arithmetic
          // (dsp/clock_recovery/clock_recovery_gardner.cpp: branched_clip on floats). One lane of the clock-recovery kernel with the Gardner iteration.
            std::vector<float> mmb;
            design::mm_bank(128, 8, mmb);
            DevBuf<float> db;
            db.reserve(mmb.size());
            SD_HIP(hipMemcpy(db.p, mmb.data(), mmb.size() * sizeof(float), hipMemcpyHostToDevice));
            MmParams p{};
            p.omega_gain = params[1];
            p.mu_gain = params[3];
            p.omega_mid = params[0];
            p.omega_limit = params[4] * params[0];
            p.init_mu = params[2];
            p.bank = db.p;
            p.cap = (int)out_cap;
            p.cg = g;
            p.rot = nullptr;
            p.order = 4;
            p.loop = 1;
            p.clip_float = kind == 10 ? 1 : 0;
            p.back = (int)std::floor(((double)p.omega_mid + std::fabs((double)p.omega_limit)) / 2.0) + 1;
            MmState s0{};
            s0.mu = params[2];
            s0.omega = params[0];
            DevBuf<MmState> st;
            st.reserve(3);
            DevBuf<int> cnt;
            cnt.reserve(2);
            SD_HIP(hipMemcpy(st.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            DevBuf<MmCert> cc;
            cc.reserve(2);
            launch_mm(X, Y, cnt.p, g, p, st.p, st.p + 1, st.p + 2, cc.p, cc.p + 1, nullptr, 0, nullptr);
            int c = 0;
            SD_HIP(hipMemcpy(&c, cnt.p, sizeof(int), hipMemcpyDeviceToHost));
            nout = c;
        }
        else if (kind == 8)
        { // PLLCarrierTrackingBlock(loop_bw, max, min), pll_carrier_tracking.cpp:8-66
            if (out_cap < n)
                throw HipError("output too small");
            PllParams p{};
            design::costas_gains(params[0], p.alpha, p.beta);
            p.fmax = params[1];
            p.fmin = params[2];
            const std::vector<float> at = design::atan_table();
            DevBuf<float> dt;
            dt.reserve(at.size());
            SD_HIP(hipMemcpy(dt.p, at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice));
            p.atan_tab = dt.p;
            CostasState s0{0, 0};
            DevBuf<CostasState> st;
            st.reserve(3);
            SD_HIP(hipMemcpy(st.p, &s0, sizeof(s0), hipMemcpyHostToDevice));
            launch_pll(X, Y, g, p, st.p, st.p + 1, st.p + 2, nullptr, 0, nullptr);
            SD_HIP(hipDeviceSynchronize()); // dt / st go out of scope below
        }
        else
            throw HipError("unknown block kind");
        SD_HIP(hipDeviceSynchronize());
        return nout;
        SD_GUARD_END(-1)
    }
}
