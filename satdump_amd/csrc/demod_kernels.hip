// demod_kernels.hip -- psk_demod DSP chain on gfx950 (see demod_kernels.h for the scheme).
//
// Arithmetic contract: every float operation rounds exactly where the reference's x86-64 -O2 build
// rounds (separate mul / add, no FMA: the library is compiled with -ffp-contract=off), sqrt is the
// IEEE correctly-rounded one, and sinf/cosf follow glibc 2.35's algorithm (sincosf.h) in double
// precision with the same polynomial and reduction, so that a single sequential lane reproduces the
// reference bit for bit (tests/test_demod_gpu.py, exact mode).
#include "demod_kernels.h"
#include <optional>

namespace sdhip
{
    // =============================================================================================
    // glibc 2.35 sinf / cosf (sysdeps/ieee754/flt-32/s_sincosf.h, x86-64 FMA build), |x| < 120
    // =============================================================================================
    struct SinCosTab
    {
        double c0, c1, c2, c3, c4, s1, s2, s3;
    };
    __device__ __forceinline__ float sd_sinf_poly(double x, double x2, bool neg_tab, int n)
    {
        // __sincosf_table[0] / [1]: table 1 negates the cosine coefficients
        const double sgn = neg_tab ? -1.0 : 1.0;
        if ((n & 1) == 0)
        {
            const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
            const double x3 = x * x2;
            const double s1 = fma(x2, s3c, s2c);
            const double x7 = x3 * x2;
            const double s = fma(x3, s1c, x);
            return (float)fma(x7, s1, s);
        }
        else
        {
            const double c0 = sgn * 0x1p0, c1c = sgn * -0x1.ffffffd0c621cp-2, c2c = sgn * 0x1.55553e1068f19p-5, c3c = sgn * -0x1.6c087e89a359dp-10,
                         c4c = sgn * 0x1.99343027bf8c3p-16;
            const double x4 = x2 * x2;
            const double c2 = fma(x2, c4c, c3c);
            const double c1 = fma(x2, c1c, c0);
            const double x6 = x4 * x2;
            const double c = fma(x4, c2c, c1);
            return (float)fma(x6, c2, c);
        }
    }
    __device__ __forceinline__ unsigned sd_abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
    __device__ __forceinline__ double sd_reduce_fast(double x, int *np)
    {
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double r = x * hpi_inv;
        const int n = ((int)r + 0x800000) >> 24;
        *np = n;
        return fma(-(double)n, hpi, x);
    }
    __device__ __forceinline__ float sd_sinf(float y)
    {
        double x = (double)y;
        if (sd_abstop12(y) < sd_abstop12(0x1.921FB6p-1f))
        {
            const double s = x * x;
            if (sd_abstop12(y) < sd_abstop12(0x1p-12f))
                return y;
            return sd_sinf_poly(x, s, false, 0);
        }
        int n;
        x = sd_reduce_fast(x, &n);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; // sign[] = {1,-1,-1,1}
        return sd_sinf_poly(x * sgn, x * x, (n & 2) != 0, n);
    }
    __device__ __forceinline__ float sd_cosf(float y)
    {
        double x = (double)y;
        if (sd_abstop12(y) < sd_abstop12(0x1.921FB6p-1f))
        {
            if (sd_abstop12(y) < sd_abstop12(0x1p-12f))
                return 1.0f;
            return sd_sinf_poly(x, x * x, false, 1);
        }
        int n;
        x = sd_reduce_fast(x, &n);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return sd_sinf_poly(x * sgn, x * x, (n & 2) != 0, n ^ 1);
    }

    // cosf(y) and sinf(y) of the SAME argument, results bit-identical to sd_cosf(y) / sd_sinf(y), without a branch: both
    // functions reduce y the same way (n, reduced x, quadrant sign, table), so one reduction serves both, the sine and the
    // cosine polynomial are each evaluated once, and the quadrant decides which result is which. The |y| < pi/4 entry of
    // glibc is the general path with n = 0 (fma(-0.0, hpi, x) == x), only |y| < 2^-12 returns y / 1.0f outright. Table 1 is
    // table 0 with the cosine coefficients negated: fma and the final conversion commute with negation, so the negated table
    // is the negated result. (The lanes of a wave sit at unrelated phases: the two-function form ran all four polynomial
    // branches for every sample, ~70 f64 and ~70 f32 VALU instructions plus ~80 scalar/branch instructions per sample.)
    __device__ __forceinline__ void sd_sincosf(float y, float &sn, float &cs)
    {
        const double x0 = (double)y;
        int n;
        const double xr = sd_reduce_fast(x0, &n);
        const double xs = ((n + 1) & 2) ? -xr : xr; // sign[] = {1,-1,-1,1} by n & 3
        const double x2 = xr * xr;
        // sine polynomial
        const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
        const double x3 = xs * x2;
        const double s1 = fma(x2, s3c, s2c);
        const double x7 = x3 * x2;
        const double sp = fma(x7, s1, fma(x3, s1c, xs));
        // cosine polynomial (table 0), negated for table 1
        const double c1c = -0x1.ffffffd0c621cp-2, c2c = 0x1.55553e1068f19p-5, c3c = -0x1.6c087e89a359dp-10, c4c = 0x1.99343027bf8c3p-16;
        const double x4 = x2 * x2;
        const double c2 = fma(x2, c4c, c3c);
        const double c1 = fma(x2, c1c, 0x1p0);
        const double x6 = x4 * x2;
        const double cp0 = fma(x6, c2, fma(x4, c2c, c1));
        const float fs = (float)sp, fc0 = (float)cp0;
        const float fc = (n & 2) ? -fc0 : fc0;
        const bool odd = (n & 1) != 0;
        const bool tiny = sd_abstop12(y) < sd_abstop12(0x1p-12f);
        sn = tiny ? y : (odd ? fc : fs);
        cs = tiny ? 1.0f : (odd ? fs : fc);
    }

    // ---- FAST arithmetic of the chunk-parallel mode (k_afc<ORDER, true>) ------------------------------------------------------------------
    // The chunk-parallel mode is held to the 1e-5 soft-symbol contract, not to bit identity (no time-parallel schedule of these loops can
    // be: DESIGN.md 2), and its float operations need not round where the reference's do. Float sine / cosine with <= 6e-8 absolute error
    // (Cody-Waite reduction by pi/2 with a fused multiply-add, Taylor polynomials of degree 9 / 8 on [-pi/4, pi/4]: truncation 1.8e-9 /
    // 2.4e-8) instead of glibc's double-precision evaluation, the hardware square root (1 ulp) instead of the correctly rounded one, fused
    // multiply-adds in the matched filter: ~60 of the fused stage's ~170 VALU instructions per sample. exact = 1 keeps every rounding.
    __device__ __forceinline__ void sd_sincosf_fast(float y, float &sn, float &cs)
    {
        const float k = rintf(y * 0.63661977236758134308f); // |y| <= 2 pi: |k| <= 4
        float r = fmaf(-k, 1.57079637050628662109375f, y);  // pi/2 rounded to float; k times it is exact inside the fused operation
        r = fmaf(-k, -4.37113900018624283e-8f, r);          // pi/2 - (float)(pi/2)
        const float r2 = r * r;
        const float sp = fmaf(r2, fmaf(r2, fmaf(r2, 2.75573192239858906526e-6f, -1.98412698412698412698e-4f), 8.33333333333333333333e-3f), -1.66666666666666666667e-1f);
        const float s = fmaf(r * r2, sp, r);
        const float cp = fmaf(r2, fmaf(r2, fmaf(r2, 2.48015873015873015873e-5f, -1.38888888888888888889e-3f), 4.16666666666666666667e-2f), -0.5f);
        const float c = fmaf(r2, cp, 1.0f);
        const int q = (int)k & 3;
        const float a = (q & 1) ? c : s, b = (q & 1) ? s : c; // sin takes (s, c, -s, -c), cos takes (c, -s, -c, s) by quadrant
        sn = (q & 2) ? -a : a;
        cs = ((q + 1) & 2) ? -b : b;
    }
    __device__ __forceinline__ float sd_sqrt_fast(float x)
    {
#ifdef SDHIP_HOST_TWIN
        return sqrtf(x);
#else
        return __builtin_amdgcn_sqrtf(x); // v_sqrt_f32, 1 ulp
#endif
    }

    // =============================================================================================
    // format conversion
    // =============================================================================================
    __global__ __launch_bounds__(256) void k_convert(const void *in, int fmt, int iq_swap, long long n, cf32 *out)
    {
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n)
            return;
        float re, im;
        if (fmt == 0)
        {
            const cf32 v = ((const cf32 *)in)[i];
            re = v.re;
            im = v.im;
        }
        else if (fmt == 1)
        { // volk_16i_s32f_convert_32f(.., 32767): x * (1.0f / 32767)
            const short2 v = ((const short2 *)in)[i];
            const float s = 1.0f / 32767.0f;
            re = (float)v.x * s;
            im = (float)v.y * s;
        }
        else if (fmt == 2)
        {
            const char2 v = ((const char2 *)in)[i];
            const float s = 1.0f / 127.0f;
            re = (float)v.x * s;
            im = (float)v.y * s;
        }
        else if (fmt == 3)
        { // cu8: (x - 127) * (1.0 / 127.0), the product taken in double and rounded once (baseband_interface.h:190-198)
            const uchar2 v = ((const uchar2 *)in)[i];
            re = (float)((double)((int)v.x - 127) * (1.0 / 127.0));
            im = (float)((double)((int)v.y - 127) * (1.0 / 127.0));
        }
        else
        { // cs32: volk_32i_s32f_convert_32f(.., 2147483647): (float)x * (1.0f / 2147483647) (baseband_interface.h:175-178)
            const int2 v = ((const int2 *)in)[i];
            const float sc = 1.0f / 2147483647.0f;
            re = (float)v.x * sc;
            im = (float)v.y * sc;
        }
        if (iq_swap)
        {
            const float t = re;
            re = im;
            im = t;
        }
        out[i].re = re;
        out[i].im = im;
    }
    void launch_convert(const void *in, int fmt, int iq_swap, long long n, cf32 *out, hipStream_t st)
    {
        if (n <= 0)
            return;
        ProfScope _ps("k_convert", st);
        hipLaunchKernelGGL(k_convert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, fmt, iq_swap, n, out);
    }

    // =============================================================================================
    // DC block (sequential; the option is off in every shipped PSK pipeline of BASELINE.json)
    // =============================================================================================
    __global__ void k_dcblock_seq(const cf32 *x, cf32 *y, long long n, DcState *state)
    {
        if (threadIdx.x != 0 || blockIdx.x != 0)
            return;
        const float alpha = 0.0001f, beta = 1.0f - 0.0001f;
        float ar = state->acc_re, ai = state->acc_im;
        for (long long i = 0; i < n; i++)
        {
            const float xr = x[i].re, xi = x[i].im;
            ar = ar * beta + xr * alpha;
            ai = ai * beta + xi * alpha;
            y[i].re = xr - ar;
            y[i].im = xi - ai;
        }
        state->acc_re = ar;
        state->acc_im = ai;
    }
    void launch_dcblock_seq(const cf32 *x, cf32 *y, long long n, DcState *state, hipStream_t st)
    {
        ProfScope _ps("k_dcblock_seq", st);
        hipLaunchKernelGGL(k_dcblock_seq, dim3(1), dim3(64), 0, st, x, y, n, state);
    }

    // =============================================================================================
    // frequency shift (demod_kernels.h)
    // =============================================================================================
    __global__ void k_rotator_seq(const cf32 *x, cf32 *y, long long n, RotState *state, float dre, float dim, int buf_len, int pos0)
    {
        if (threadIdx.x != 0 || blockIdx.x != 0)
            return;
        float pr = state->re, pi = state->im;
        auto renorm = [&]() { // phase /= hypotf(re, im): glibc's hypotf takes the root in double and narrows
            const float h = (float)sqrt((double)pr * (double)pr + (double)pi * (double)pi);
            pr = pr / h;
            pi = pi / h;
        };
        int pos = pos0; // samples of the current call consumed so far
        for (long long i = 0; i < n; i++)
        {
            const float xr = x[i].re, xi = x[i].im;
            y[i].re = xr * pr - xi * pi;
            y[i].im = xr * pi + xi * pr;
            const float nr = pr * dre - pi * dim, ni = pr * dim + pi * dre;
            pr = nr;
            pi = ni;
            pos++;
            if (pos == buf_len)
            { // end of a call: a 512-sample run that just completed has been renormalised by the run loop, a remainder by the tail rule
                renorm();
                pos = 0;
            }
            else if ((pos & 511) == 0)
                renorm();
        }
        state->re = pr;
        state->im = pi;
    }
    void launch_rotator_seq(const cf32 *x, cf32 *y, long long n, RotState *state, float dre, float dim, int buf_len, int pos0, hipStream_t st)
    {
        ProfScope _ps("k_rotator_seq", st);
        hipLaunchKernelGGL(k_rotator_seq, dim3(1), dim3(64), 0, st, x, y, n, state, dre, dim, buf_len, pos0);
    }
    __global__ __launch_bounds__(256) void k_rotator_par(const cf32 *x, cf32 *y, long long n, long long abs0, unsigned long long f_fix, float mag_eps, int buf_len)
    {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= n)
            return;
        const unsigned long long a = (unsigned long long)(abs0 + i);
        const unsigned long long turns = a * f_fix;                                       // phase in 2^-64 turns, exact modulo one turn
        const float ang = (float)(int)(unsigned)(turns >> 32) * 1.46291807926715968105e-9f; // * 2 pi / 2^32: [-pi, pi)
        float sn, cs;
        sd_sincosf_fast(ang, sn, cs);
        const int run = (int)(a % (unsigned long long)buf_len) & 511; // steps since the last renormalisation
        const float m = 1.0f + (float)run * mag_eps;
        cs *= m;
        sn *= m;
        const cf32 v = x[i];
        y[i] = cf32{v.re * cs - v.im * sn, v.re * sn + v.im * cs};
    }
    void launch_rotator_par(const cf32 *x, cf32 *y, long long n, long long abs0, unsigned long long f_fix, float mag_eps, int buf_len, hipStream_t st)
    {
        if (n <= 0)
            return;
        ProfScope _ps("k_rotator_par", st);
        hipLaunchKernelGGL(k_rotator_par, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, abs0, f_fix, mag_eps, buf_len);
    }

    // =============================================================================================
    // Doppler correction (demod_kernels.h)
    // =============================================================================================
    __global__ void k_doppler_seq(const cf32 *x, cf32 *y, long long n, DopState *state, float alpha, float target_cur, const float *targets, int buf_len, int pos0)
    {
        if (threadIdx.x != 0 || blockIdx.x != 0)
            return;
        float phase = state->phase, freq = state->freq, targ = target_cur;
        int pos = pos0, tk = 0;
        for (long long i = 0; i < n; i++)
        {
            if (pos == buf_len)
            { // a new source buffer: the block has recomputed its target behind the previous one
                targ = targets[tk++];
                pos = 0;
            }
            float sn, cs;
            sd_sincosf(-phase, sn, cs);
            const float xr = x[i].re, xi = x[i].im;
            y[i].re = (xr * cs) - (xi * sn); // complex_t * complex_t(cosf(-phase), sinf(-phase))
            y[i].im = (xi * cs) + (xr * sn);
            phase = phase + freq;
            while ((double)phase > 2.0 * 3.14159265358979323846)
                phase = (float)((double)phase - 2.0 * 3.14159265358979323846);
            while ((double)phase < -2.0 * 3.14159265358979323846)
                phase = (float)((double)phase + 2.0 * 3.14159265358979323846);
            // curr_freq * (1.0 - d_alpha) + targ_freq * d_alpha: the first product in double, the second in float, the sum in double
            freq = (float)((double)freq * (1.0 - (double)alpha) + (double)(targ * alpha));
            pos++;
        }
        state->phase = phase;
        state->freq = freq;
    }
    void launch_doppler_seq(const cf32 *x, cf32 *y, long long n, DopState *state, float alpha, float target_cur, const float *targets, int buf_len, int pos0, hipStream_t st)
    {
        ProfScope _ps("k_doppler_seq", st);
        hipLaunchKernelGGL(k_doppler_seq, dim3(1), dim3(64), 0, st, x, y, n, state, alpha, target_cur, targets, buf_len, pos0);
    }
    __global__ __launch_bounds__(256) void k_doppler_par(const cf32 *x, cf32 *y, long long n, const DopStart *starts, double alpha, double log_beta, int buf_len, int pos0)
    {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= n)
            return;
        const long long a = i + pos0;
        const int b = (int)(a / buf_len);
        const double k = (double)(a - (long long)b * buf_len) - (b == 0 ? (double)pos0 : 0.0); // steps since this buffer's start state was taken
        const DopStart s = starts[b];
        const double bk = exp(log_beta * k);
        double ph = s.ph0 + k * s.target + (s.f0 - s.target) * (1.0 - bk) / alpha;
        ph -= 6.283185307179586476925 * rint(ph / 6.283185307179586476925);
        float sn, cs;
        sd_sincosf_fast((float)-ph, sn, cs);
        const cf32 v = x[i];
        y[i] = cf32{v.re * cs - v.im * sn, v.im * cs + v.re * sn};
    }
    void launch_doppler_par(const cf32 *x, cf32 *y, long long n, const DopStart *starts, double alpha, int buf_len, int pos0, hipStream_t st)
    {
        if (n <= 0)
            return;
        ProfScope _ps("k_doppler_par", st);
        hipLaunchKernelGGL(k_doppler_par, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, starts, alpha, log1p(-alpha), buf_len, pos0);
    }

    // =============================================================================================
    // rational resampler: output m uses inputs [inc-(nt-1), inc] with phase ctr (sequential-order dot product)
    // =============================================================================================
    typedef float v2f __attribute__((ext_vector_type(2))); // (re, im) pair: mul / add map to v_pk_mul_f32 / v_pk_add_f32

    constexpr int RS_BLOCK = 256, RS_PER = 4;        // outputs per block = RS_BLOCK * RS_PER (thread t: m0 + t + RS_BLOCK*r)
    constexpr int RS_MAX_BANK = 4096;                // floats of polyphase bank kept in LDS
    constexpr int RS_MAX_TILE = 2560;                // input samples staged per block
    // Block = 1024 consecutive outputs. Their input span (~1024*decim/interp + ntaps samples) and the whole polyphase bank
    // are staged in LDS with coalesced loads; every thread then accumulates 4 independent outputs tap by tap, oldest
    // sample first, mul and add rounded separately exactly like the scalar reference loop (rational_resampler.cpp:49-56 ->
    // volk generic dot product).
    // x may be the caller's own cf32 buffer (read in place): samples at negative indices come from `hist` (the DEMOD_HIST
    // samples preceding x[0]) and nothing at or past nin is touched.
    __global__ __launch_bounds__(RS_BLOCK) void k_resample(const cf32 *x, const cf32 *hist, long long nin, ResampParams p, int ctr0, int inc0, cf32 *y,
                                                           long long nout)
    {
        __shared__ float bank[RS_MAX_BANK];
        __shared__ v2f tile[RS_MAX_TILE];
        const long long m0 = (long long)blockIdx.x * (RS_BLOCK * RS_PER);
        const int nb = p.interp * p.ntaps;
        for (int i = (int)threadIdx.x; i < nb; i += RS_BLOCK)
            bank[i] = p.bank[i];
        long long mlast = m0 + RS_BLOCK * RS_PER - 1;
        if (mlast >= nout)
            mlast = nout - 1;
        const long long first = inc0 + ((long long)ctr0 + m0 * p.decim) / p.interp - (p.ntaps - 1);
        const long long last = inc0 + ((long long)ctr0 + mlast * p.decim) / p.interp;
        const int span = (int)(last - first + 1);
        for (int i = (int)threadIdx.x; i < span; i += RS_BLOCK)
        {
            const long long idx = first + i;
            const cf32 v = (idx < 0) ? hist[DEMOD_HIST + idx] : (idx < nin ? x[idx] : cf32{0.0f, 0.0f});
            tile[i] = v2f{v.re, v.im};
        }
        __syncthreads();
        v2f acc[RS_PER];
        int off[RS_PER], row[RS_PER];
#pragma unroll
        for (int r = 0; r < RS_PER; r++)
        {
            long long m = m0 + (int)threadIdx.x + RS_BLOCK * r;
            if (m >= nout)
                m = nout - 1; // clamp: computed, not stored
            const long long ph = (long long)ctr0 + m * p.decim;
            off[r] = (int)(inc0 + ph / p.interp - (p.ntaps - 1) - first);
            row[r] = (int)(ph % p.interp) * p.ntaps;
            acc[r] = v2f{0.0f, 0.0f};
        }
        for (int k = 0; k < p.ntaps; k++)
        {
#pragma unroll
            for (int r = 0; r < RS_PER; r++)
            {
                const float tk = bank[row[r] + k];
                const v2f prod = tile[off[r] + k] * v2f{tk, tk};
                acc[r] = acc[r] + prod;
            }
        }
#pragma unroll
        for (int r = 0; r < RS_PER; r++)
        {
            const long long m = m0 + (int)threadIdx.x + RS_BLOCK * r;
            if (m < nout)
                reinterpret_cast<v2f *>(y)[m] = acc[r];
        }
    }
    struct ResampIndex
    {
        long long o_first, inc0; // first window-end offset of the launch; carried input index of the stream
        long long qb0, qi0;      // output index / input-index quotient belonging to o_first
        int rb, ri;              // and their remainders (constant over the launch)
    };
    // Decimating ratios close to 1 (GOES: 9/10): consecutive outputs start 1 or 2 input samples apart, so the 32 lanes of an
    // LDS access group span more than 32 tile slots and every tile read of k_resample takes two passes (SQ: bank-conflict
    // cycles ~3x the active LDS cycles). Here a lane owns an INPUT OFFSET instead: the one output (if any) whose window ends
    // at that sample. Lanes then read stride-1 tile slots -- conflict free -- at the price of decim-interp idle lanes in decim.
    // Same arithmetic per output (taps oldest first, mul and add rounded separately).
    // A thread's RS_PER offsets are `stride` = (256 / decim) * decim apart: same polyphase arm (the hit/miss pattern and the
    // arm repeat every decim offsets), so the tap is read once per k for all of them.
    // Index arithmetic: offset o is the window end of output m iff inc0 + (ctr0 + m*decim)/interp == o with
    // m = ceil(((o - inc0)*interp - ctr0) / decim). Done naively that is three 64-bit divisions per output (~2000 VALU
    // instructions per thread against ~500 useful ones: the first version of this kernel ran at 72 % VALU issue doing mostly
    // that). A block covers RS_PER*stride offsets = a whole number of decim-periods, so the 64-bit quotients advance by a
    // constant per block and per r; the host divides once (ResampIndex), a thread is left with three small 32-bit divisions.
    __global__ __launch_bounds__(RS_BLOCK) void k_resample_byoffset(const cf32 *x, const cf32 *hist, long long nin, ResampParams p, ResampIndex ix, cf32 *y,
                                                                    long long nout, int stride)
    {
        __shared__ float bank[RS_MAX_BANK];
        __shared__ v2f tile[RS_BLOCK * RS_PER + 64];
        const int nb = p.interp * p.ntaps;
        for (int i = (int)threadIdx.x; i < nb; i += RS_BLOCK)
            bank[i] = p.bank[i];
        const long long blk = (long long)blockIdx.x;
        const long long o_base = ix.o_first + blk * (stride * RS_PER); // window-end offsets [o_base, o_base + RS_PER*stride)
        const long long first = o_base - (p.ntaps - 1);
        const int span = stride * RS_PER + p.ntaps - 1;
        for (int i = (int)threadIdx.x; i < span; i += RS_BLOCK)
        {
            const long long idx = first + i;
            const cf32 v = (idx < 0) ? hist[DEMOD_HIST + idx] : (idx < nin ? x[idx] : cf32{0.0f, 0.0f});
            tile[i] = v2f{v.re, v.im};
        }
        __syncthreads();
        const int t0 = (int)threadIdx.x;
        if (t0 >= stride)
            return;
        const int per_dec = stride / p.decim;                      // decim-periods per stride
        const long long qb = ix.qb0 + blk * (long long)(RS_PER * per_dec * p.interp); // outputs in front of this block's first candidate
        const long long qi = ix.qi0 + blk * (long long)(RS_PER * stride);
        const unsigned u = (unsigned)ix.rb + (unsigned)t0 * (unsigned)p.interp;
        const unsigned mv0 = u / (unsigned)p.decim;
        const unsigned t2 = (unsigned)ix.ri + mv0 * (unsigned)p.decim;
        const unsigned a0 = t2 / (unsigned)p.interp, arm = t2 - a0 * (unsigned)p.interp;
        const bool hit0 = ix.inc0 + qi + (long long)a0 == o_base + t0; // the same for every r (offsets a whole number of periods apart)
        const int row = (int)arm * p.ntaps;
        v2f acc[RS_PER];
        long long mm[RS_PER];
#pragma unroll
        for (int r = 0; r < RS_PER; r++)
        {
            const long long m = qb + (long long)mv0 + (long long)r * (per_dec * p.interp);
            mm[r] = (hit0 && m < nout) ? m : -1;
            acc[r] = v2f{0.0f, 0.0f};
        }
        const float *brow = bank + row;
        const v2f *trow = tile + t0;
        for (int k = 0; k < p.ntaps; k++)
        {
            const float tk = brow[k];
            const v2f tt{tk, tk};
#pragma unroll
            for (int r = 0; r < RS_PER; r++)
            {
                const v2f prod = trow[stride * r + k] * tt;
                acc[r] = acc[r] + prod;
            }
        }
#pragma unroll
        for (int r = 0; r < RS_PER; r++)
            if (mm[r] >= 0)
                reinterpret_cast<v2f *>(y)[mm[r]] = acc[r];
    }

    // Register-window variant for a ratio known at compile time (GOES HRIT: 9/10, 38 taps per arm). k_resample_byoffset reads
    // every sample of every window from LDS (ntaps * 8 B per output: the LDS pipe, not HBM, bounds it). Here a thread owns one
    // whole decim-period starting at an output of arm 0: D consecutive window-end offsets -> I outputs whose (offset, arm)
    // pattern (j*D/I, j*D%I) is the same for every thread. It pulls its NT + EMAX samples into registers once (8x fewer LDS
    // bytes per output for 9/10), the arm of output j is wave-uniform so the taps arrive through scalar loads, and the
    // outputs go back through LDS so that the global stores are contiguous. Accumulation order per output is unchanged.
    constexpr int RSP_BLOCK = 256;
    // grid of a persistent kernel: exactly the blocks that are resident at once (CUs x occupancy), see k_resample_period
    template <class K>
    static int resident_grid(K kernel, int block)
    {
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block, 0) != hipSuccess || occ < 1)
            occ = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        return occ * cus;
    }
    // Persistent blocks walk over the tiles with stride gridDim.x and fetch the NEXT tile's samples into registers before
    // they compute the current one: with one tile per block the loads of a block only overlap other blocks' compute, and
    // at 3-4 resident blocks per CU that left HBM at ~3.5 TB/s.
    template <int I, int D, int NT>
    __global__ __launch_bounds__(RSP_BLOCK) void k_resample_period(const cf32 *x, const cf32 *hist, long long nin, const float *__restrict__ bank,
                                                                    long long off0, long long m_start, cf32 *y, long long nout, long long ntiles)
    {
        constexpr int EMAX = ((I - 1) * D) / I;   // largest window-end offset inside a period
        constexpr int WIN = NT + EMAX;            // samples one thread touches
        constexpr int SPAN = (RSP_BLOCK - 1) * D + WIN;
        constexpr int LOADS = (SPAN + RSP_BLOCK - 1) / RSP_BLOCK;
        static_assert(RSP_BLOCK * I <= SPAN, "output staging reuses the input tile");
        __shared__ v2f tile[SPAN];
        const int t = (int)threadIdx.x;
        v2f pre[LOADS];
        unsigned ok = 0; // bit l: pre[l] is a real sample (not clamped)
        auto fetch = [&](long long blk) {
            ok = 0;
            const long long first = off0 + blk * (RSP_BLOCK * D) - (NT - 1); // input index of tile[0]
#pragma unroll
            for (int l = 0; l < LOADS; l++)
            {
                // branch-free: always load from a clamped (valid) address, zero what lies outside [-DEMOD_HIST, nin)
                const long long idx = first + t + RSP_BLOCK * l;
                long long c = idx < nin - 1 ? idx : nin - 1;
                c = c > -DEMOD_HIST ? c : -DEMOD_HIST;
                const v2f *src = c >= 0 ? reinterpret_cast<const v2f *>(x) + c : reinterpret_cast<const v2f *>(hist) + (DEMOD_HIST + c);
                pre[l] = *src; // masked when it is written to LDS: a select here would make the compiler wait for each load in turn
                ok |= (c == idx ? 1u : 0u) << l;
            }
        };
        long long blk = (long long)blockIdx.x;
        if (blk < ntiles)
            fetch(blk);
        for (; blk < ntiles; blk += gridDim.x)
        {
#pragma unroll
            for (int l = 0; l < LOADS; l++)
                if (t + RSP_BLOCK * l < SPAN)
                    tile[t + RSP_BLOCK * l] = ((ok >> l) & 1u) ? pre[l] : v2f{0.0f, 0.0f};
            __syncthreads();
            v2f s[WIN];
#pragma unroll
            for (int i = 0; i < WIN; i++)
                s[i] = tile[t * D + i];
            __syncthreads();
            if (blk + gridDim.x < ntiles)
                fetch(blk + gridDim.x);
            int z = 0;
            asm volatile("" : "+s"(z)); // opaque zero: keeps the (loop-invariant) scalar tap loads inside the tile loop; hoisted they spill ~340 SGPRs
#pragma unroll
            for (int j = 0; j < I; j++)
            {
                const int e = (j * D) / I, a = (j * D) % I;
                v2f acc{0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < NT; k++)
                {
                    const float tk = bank[z + a * NT + k];
                    const v2f prod = s[e + k] * v2f{tk, tk};
                    acc = acc + prod;
                }
                tile[t * I + j] = acc;
            }
            __syncthreads();
            const long long mb = m_start + blk * (RSP_BLOCK * I); // output index of tile[0]
            for (int i = t; i < RSP_BLOCK * I; i += RSP_BLOCK)
            {
                const long long m = mb + i;
                if (m >= 0 && m < nout)
                    reinterpret_cast<v2f *>(y)[m] = tile[i];
            }
            __syncthreads();
        }
    }
    static bool resample_period_enabled()
    {
        static const bool on = [] {
            const char *e = getenv("SDHIP_RESAMP_PERIOD");
            return !(e && e[0] == '0');
        }();
        return on;
    }

    // fallback for banks / spans that do not fit the LDS budget (very large interpolation factors)
    __global__ __launch_bounds__(256) void k_resample_big(const cf32 *x, const cf32 *hist, long long nin, ResampParams p, int ctr0, int inc0, cf32 *y,
                                                           long long nout)
    {
        const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (m >= nout)
            return;
        const long long ph = (long long)ctr0 + m * p.decim;
        const long long inc = inc0 + ph / p.interp;
        const int ctr = (int)(ph % p.interp);
        const float *t = p.bank + (size_t)ctr * p.ntaps;
        const long long b = inc - (p.ntaps - 1);
        float re = 0.0f, im = 0.0f;
        for (int k = 0; k < p.ntaps; k++)
        {
            const long long idx = b + k;
            const cf32 v = (idx < 0) ? hist[DEMOD_HIST + idx] : (idx < nin ? x[idx] : cf32{0.0f, 0.0f});
            const float tk = t[k];
            re = re + v.re * tk;
            im = im + v.im * tk;
        }
        y[m].re = re;
        y[m].im = im;
    }
    void launch_resample(const cf32 *x, const cf32 *hist, long long nin, const ResampParams &p, int ctr0, int inc0, cf32 *y, long long nout, hipStream_t st)
    {
        if (nout <= 0)
            return;

        const long long span_max = ((long long)RS_BLOCK * RS_PER * p.decim) / p.interp + p.ntaps + 2;
        if (p.interp == 9 && p.decim == 10 && p.ntaps == 38 && resample_period_enabled())
        {
            // first output whose arm is 0, then one period back so that outputs 0 .. m_first-1 are covered (m < 0 is masked)
            long long m_first = 0;
            while (((long long)ctr0 + m_first * p.decim) % p.interp != 0)
                m_first++;
            const long long m_start = m_first - p.interp;
            const long long off0 = inc0 + ((long long)ctr0 + m_start * p.decim) / p.interp; // exact division
            const long long nper = (nout - m_start + p.interp - 1) / p.interp;
            ProfScope _ps("k_resample_period", st);
            const long long ntiles = (nper + RSP_BLOCK - 1) / RSP_BLOCK;
            static const int grid = resident_grid(k_resample_period<9, 10, 38>, RSP_BLOCK);
            hipLaunchKernelGGL((k_resample_period<9, 10, 38>), dim3((unsigned)(ntiles < grid ? ntiles : grid)), dim3(RSP_BLOCK), 0, st, x, hist, nin,
                               p.bank, off0, m_start, y, nout, ntiles);
        }
        else if (p.interp * p.ntaps <= RS_MAX_BANK && p.ntaps <= 64 && p.decim > p.interp && 4 * p.decim <= 5 * p.interp && p.decim <= 64)
        {
            const long long o_first = inc0 + (long long)ctr0 / p.interp;                              // window end of output 0
            const long long o_last = inc0 + ((long long)ctr0 + (nout - 1) * (long long)p.decim) / p.interp; // ... of the last output
            const int stride = (RS_BLOCK / p.decim) * p.decim;
            // m(o) = floor((N0 + (o - o_first)*interp) / decim), N0 = (o_first - inc0)*interp - ctr0 + decim - 1 >= 0
            ResampIndex ix;
            ix.o_first = o_first;
            ix.inc0 = inc0;
            const long long N0 = (o_first - inc0) * (long long)p.interp - ctr0 + p.decim - 1;
            ix.qb0 = N0 / p.decim;
            ix.rb = (int)(N0 % p.decim);
            const long long PH0 = (long long)ctr0 + ix.qb0 * p.decim; // phase counter of output qb0
            ix.qi0 = PH0 / p.interp;
            ix.ri = (int)(PH0 % p.interp);
            ProfScope _ps("k_resample_byoffset", st);
            hipLaunchKernelGGL(k_resample_byoffset, dim3((unsigned)((o_last - o_first + stride * RS_PER) / (stride * RS_PER))), dim3(RS_BLOCK), 0, st, x, hist, nin, p,
                               ix, y, nout, stride);
        }
        else if (p.interp * p.ntaps <= RS_MAX_BANK && span_max <= RS_MAX_TILE)
        {
            ProfScope _ps("k_resample", st);
            hipLaunchKernelGGL(k_resample, dim3((unsigned)((nout + RS_BLOCK * RS_PER - 1) / (RS_BLOCK * RS_PER))), dim3(RS_BLOCK), 0, st, x, hist, nin, p, ctr0, inc0, y, nout);
        }
        else
        {
            ProfScope _ps("k_resample_big", st);
            hipLaunchKernelGGL(k_resample_big, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, x, hist, nin, p, ctr0, inc0, y, nout);
        }
    }

    // =============================================================================================
    // FIR: y[i] = sum_j x[i-(nt-1)+j] * rtaps[j], accumulated oldest sample first (volk generic order)
    // =============================================================================================
    // =============================================================================================
    // decimating FIR (power-of-two pre-decimator stages): thread per output, taps (<= 1024, reversed) in LDS. Wideband
    // recordings only; the bench configurations never come here.
    // =============================================================================================
    __global__ __launch_bounds__(256) void k_decim_fir(const cf32 *x, const cf32 *hist, long long nin, const float *__restrict__ rtaps, int ntaps, int decim,
                                                        int inc0, cf32 *y, long long nout)
    {
        __shared__ float taps[1024];
        for (int i = (int)threadIdx.x; i < ntaps; i += 256)
            taps[i] = rtaps[i];
        __syncthreads();
        const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
        if (m >= nout)
            return;
        const long long first = (long long)inc0 + m * decim - (ntaps - 1);
        float re = 0.0f, im = 0.0f;
        for (int j = 0; j < ntaps; j++)
        {
            const long long i = first + j;
            const cf32 v = i >= 0 ? x[i] : hist[ntaps + i];
            re = re + v.re * taps[j];
            im = im + v.im * taps[j];
        }
        y[m] = cf32{re, im};
    }
    void launch_decim_fir(const cf32 *x, const cf32 *hist, long long nin, const float *rtaps_dev, int ntaps, int decim, int inc0, cf32 *y, long long nout,
                          hipStream_t st)
    {
        if (nout <= 0)
            return;
        ProfScope _ps("k_decim_fir", st);
        hipLaunchKernelGGL(k_decim_fir, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, x, hist, nin, rtaps_dev, ntaps, decim, inc0, y, nout);
    }

    constexpr int FIR_MAX_TAPS = 384;
    constexpr int FIR_BLOCK = 256, FIR_PER = 4; // outputs per block = 1024 (thread t: i0 + t + 256*r)
    __global__ __launch_bounds__(FIR_BLOCK) void k_fir(const cf32 *x, cf32 *y, long long n, const float *__restrict__ rtaps, int ntaps)
    {
        __shared__ v2f tile[FIR_BLOCK * FIR_PER + FIR_MAX_TAPS];
        const long long i0 = (long long)blockIdx.x * (FIR_BLOCK * FIR_PER);
        long long cnt = n - i0;
        if (cnt > FIR_BLOCK * FIR_PER)
            cnt = FIR_BLOCK * FIR_PER;
        const int span = (int)cnt + ntaps - 1;
        const v2f *xs = reinterpret_cast<const v2f *>(x) + (i0 - (ntaps - 1));
        for (int i = (int)threadIdx.x; i < span; i += FIR_BLOCK)
            tile[i] = xs[i];
        __syncthreads();
        v2f acc[FIR_PER];
#pragma unroll
        for (int r = 0; r < FIR_PER; r++)
            acc[r] = v2f{0.0f, 0.0f};
        const int t0 = (int)threadIdx.x;
        for (int j = 0; j < ntaps; j++)
        {
            const float t = rtaps[j]; // wave-uniform: scalar load
            const v2f tt{t, t};
#pragma unroll
            for (int r = 0; r < FIR_PER; r++)
            {
                // rows past the end of a ragged last block read stale LDS: computed, never stored
                const v2f prod = tile[t0 + FIR_BLOCK * r + j] * tt;
                acc[r] = acc[r] + prod;
            }
        }
#pragma unroll
        for (int r = 0; r < FIR_PER; r++)
        {
            const long long i = i0 + t0 + FIR_BLOCK * r;
            if (i < n)
                reinterpret_cast<v2f *>(y)[i] = acc[r];
        }
    }
    // Register-window variant for a tap count known at compile time (the 31-tap RRC every pipeline of the path uses): a thread
    // pulls NT + R - 1 samples into registers once and produces R consecutive outputs from them (k_fir reads every sample of
    // every window from LDS: NT * 8 B per output, enough to keep the LDS pipe as busy as HBM). R = 10: the 16-byte LDS reads of
    // 16 consecutive lanes (20 dwords apart) fall on 16 distinct bank quads. Outputs return through LDS for contiguous stores.
    // Same accumulation order per output as k_fir.
    constexpr int FIRW_BLOCK = 256, FIRW_R = 10;
    template <int NT>
    __global__ __launch_bounds__(FIRW_BLOCK) void k_fir_window(const cf32 *x, cf32 *y, long long n, const float *__restrict__ rtaps, long long ntiles)
    {
        constexpr int OUTS = FIRW_BLOCK * FIRW_R;
        constexpr int SPAN = OUTS + NT - 1;
        constexpr int LOADS = (SPAN + FIRW_BLOCK - 1) / FIRW_BLOCK;
        __shared__ v2f tile[SPAN];
        const int t = (int)threadIdx.x;
        v2f pre[LOADS];
        unsigned ok = 0;
        auto fetch = [&](long long blk) {
            ok = 0;
            const long long i0 = blk * OUTS;
            const v2f *xs = reinterpret_cast<const v2f *>(x) + (i0 - (NT - 1));
            const long long lim = n - i0 + (NT - 1); // tile entries backed by input samples
#pragma unroll
            for (int l = 0; l < LOADS; l++)
            {
                const int i = t + FIRW_BLOCK * l; // branch-free: clamped address, zero past the end of the input
                const long long c = i < lim - 1 ? i : lim - 1;
                pre[l] = xs[c]; // masked when it is written to LDS (see k_resample_period)
                ok |= (c == i ? 1u : 0u) << l;
            }
        };
        long long blk = (long long)blockIdx.x;
        if (blk < ntiles)
            fetch(blk); // persistent blocks, next tile prefetched during compute: see k_resample_period
        for (; blk < ntiles; blk += gridDim.x)
        {
#pragma unroll
            for (int l = 0; l < LOADS; l++)
                if (t + FIRW_BLOCK * l < SPAN)
                    tile[t + FIRW_BLOCK * l] = ((ok >> l) & 1u) ? pre[l] : v2f{0.0f, 0.0f};
            __syncthreads();
            v2f s[NT + FIRW_R - 1];
#pragma unroll
            for (int i = 0; i < NT + FIRW_R - 1; i++)
                s[i] = tile[t * FIRW_R + i];
            __syncthreads();
            if (blk + gridDim.x < ntiles)
                fetch(blk + gridDim.x);
            v2f acc[FIRW_R];
#pragma unroll
            for (int r = 0; r < FIRW_R; r++)
                acc[r] = v2f{0.0f, 0.0f};
            int z = 0;
            asm volatile("" : "+s"(z)); // opaque zero: keeps the scalar tap loads inside the tile loop
#pragma unroll
            for (int j = 0; j < NT; j++)
            {
                const float tk = rtaps[z + j]; // wave-uniform: scalar load
                const v2f tt{tk, tk};
#pragma unroll
                for (int r = 0; r < FIRW_R; r++)
                {
                    const v2f prod = s[r + j] * tt;
                    acc[r] = acc[r] + prod;
                }
            }
#pragma unroll
            for (int r = 0; r < FIRW_R; r++)
                tile[t * FIRW_R + r] = acc[r];
            __syncthreads();
            const long long i0 = blk * OUTS;
            for (int i = t; i < OUTS; i += FIRW_BLOCK)
                if (i0 + i < n)
                    reinterpret_cast<v2f *>(y)[i0 + i] = tile[i];
            __syncthreads();
        }
    }
    static bool fir_window_enabled()
    {
        static const bool on = [] {
            const char *e = getenv("SDHIP_FIR_WINDOW");
            return !(e && e[0] == '0');
        }();
        return on;
    }
    void launch_fir(const cf32 *x, cf32 *y, long long n, const float *rtaps_dev, int ntaps, hipStream_t st)
    {
        if (n <= 0)
            return;
        if (ntaps == 31 && fir_window_enabled())
        {
            ProfScope _ps("k_fir_window", st);
            const long long ntiles = (n + FIRW_BLOCK * FIRW_R - 1) / (FIRW_BLOCK * FIRW_R);
            static const int grid = resident_grid(k_fir_window<31>, FIRW_BLOCK);
            hipLaunchKernelGGL((k_fir_window<31>), dim3((unsigned)(ntiles < grid ? ntiles : grid)), dim3(FIRW_BLOCK), 0, st, x, y, n, rtaps_dev, ntiles);
            return;
        }
        ProfScope _ps("k_fir", st);
        hipLaunchKernelGGL(k_fir, dim3((unsigned)((n + FIR_BLOCK * FIR_PER - 1) / (FIR_BLOCK * FIR_PER))), dim3(FIR_BLOCK), 0, st, x, y, n, rtaps_dev, ntaps);
    }

    // =============================================================================================
    // chunk-speculative driver
    // =============================================================================================
    // Stage concept:  State init(P);  void step(State&, P, x, y, i, bool write)
    // one 64-byte block of a lane's stream: 8 samples = 4 x float4
    struct Blk8
    {
        float4 a, b, c, d;
    };
    template <int D, bool FASTQ = false>
    struct AgcStageT
    {
        using P = AgcParams;
        using S = AgcState;
        static constexpr int DEPTH = D; // blocks per load group (D * 64 bytes); two groups in flight
        __device__ static __forceinline__ S init(const P &p, int k) { return S{p.starts ? p.starts[k] : p.init_gain}; } // (the scan's value at this chunk's start, if there is one)
        // early exit of a re-run lane (CKPT): is state a on the trajectory that left checkpoint b? (the AGC certificate's own rule)
        __device__ static __forceinline__ bool close(const S &a, const S &b, float tol_a, float) { return fabsf(a.gain - b.gain) <= tol_a * fabsf(b.gain); }
        __device__ static __forceinline__ void prewarm(S &, const P &, const cf32 *, long long) {}
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v) { return step_t<FASTQ>(s, p, v); }
        template <bool FAST>
        __device__ static __forceinline__ cf32 step_t(S &s, const P &p, const cf32 v)
        {
            // AGCBlock<complex_t>::work, agc.cpp:25-39
            const float ore = v.re * s.gain;
            const float oim = v.im * s.gain;
            float mag;
            if (p.input_mag)
            { // AGCFastBlock<complex_t>::process, agc_fast.cpp:37-55: mag_buf[i] = sqrtf(re * re + im * im) of the input (VOLK's generic kernel), times the gain
                float mi;
                if constexpr (FAST)
                    mi = sd_sqrt_fast(fmaf(v.re, v.re, v.im * v.im));
                else
                    mi = sqrtf(v.re * v.re + v.im * v.im);
                mag = mi * s.gain;
            }
            else if constexpr (FAST)
                mag = sd_sqrt_fast(fmaf(ore, ore, oim * oim));
            else
                mag = sqrtf(ore * ore + oim * oim) /* correctly rounded (default -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is the native approximation */;
            s.gain = s.gain + p.rate * (p.reference - mag);
            if (p.max_gain > 0.0f && s.gain > p.max_gain)
                s.gain = p.max_gain;
            return cf32{ore, oim};
        }
    };

    using AgcStage = AgcStageT<4>;

    // ---- AGC and the 31-tap RRC filter behind it as ONE lane-per-chunk stage ------------------------------------------------------
    // The filter is feed-forward, but run as its own kernel it costs a full write + read of the stream (16 B per sample of the 60 the
    // whole demodulator moves) while the AGC lanes -- one wave per SIMD, bound by their chunk-strided HBM reads -- leave the VALU
    // idle. Here the lane that produces the AGC samples of a chunk also filters them: its state carries the last NT - 1 AGC outputs,
    // every 8-sample block appends 8 new ones and emits the 8 filter outputs that end on them (same products, same ascending-tap
    // accumulation as k_fir / k_fir_window: bit for bit the same floats). The AGC output itself never goes to memory.
    // Certificate: the window of a warm-up equals the predecessor's iff the two gain trajectories had merged >= NT - 1 samples in
    // front of the boundary, so the state also remembers the gain four blocks (32 samples) ago and the verdict compares both.
    // a wave-uniform float copied into a vector register: keeps a table of loop-invariant scalars out of the scalar register file
    // (31 tap pairs there spill; packed-FP32 operands take a vector register's low half for both lanes, a scalar pair they do not)
    __device__ __forceinline__ float sd_to_vgpr(float v)
    {
        float r;
        asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
        return r;
    }
    template <bool FAST>
    struct AgcFirStageT
    {
        using P = AgcFirParams;
        using S = AgcFirState;
        static constexpr int DEPTH = 4;
        static constexpr int NT = AGCFIR_NT;
        __device__ static __forceinline__ S init(const P &p, int)
        {
            S s;
            s.gain = p.agc.init_gain;
#pragma unroll
            for (int i = 0; i < 4; i++)
                s.lag[i] = p.agc.init_gain;
#pragma unroll
            for (int i = 0; i < 2 * (NT - 1); i++)
                s.w[i] = 0.0f;
            return s;
        }
        __device__ static __forceinline__ bool close(const S &a, const S &b, float tol_a, float) { return fabsf(a.gain - b.gain) <= tol_a * fabsf(b.gain); }
        __device__ static __forceinline__ void prewarm(S &, const P &, const cf32 *, long long) {}
        __device__ static __forceinline__ v2f agc(S &s, const P &p, float re, float im)
        {
            AgcState a{s.gain};
            const cf32 o = AgcStageT<4>::template step_t<FAST>(a, p.agc, cf32{re, im});
            s.gain = a.gain;
            return v2f{o.re, o.im};
        }
        __device__ static __forceinline__ Blk8 block(S &s, const P &p, const Blk8 &c, bool write)
        {
            v2f loc[NT - 1 + 8];
#pragma unroll
            for (int i = 0; i < NT - 1; i++)
                loc[i] = v2f{s.w[2 * i], s.w[2 * i + 1]};
            const float g_in = s.gain;
            loc[NT - 1 + 0] = agc(s, p, c.a.x, c.a.y);
            loc[NT - 1 + 1] = agc(s, p, c.a.z, c.a.w);
            loc[NT - 1 + 2] = agc(s, p, c.b.x, c.b.y);
            loc[NT - 1 + 3] = agc(s, p, c.b.z, c.b.w);
            loc[NT - 1 + 4] = agc(s, p, c.c.x, c.c.y);
            loc[NT - 1 + 5] = agc(s, p, c.c.z, c.c.w);
            loc[NT - 1 + 6] = agc(s, p, c.d.x, c.d.y);
            loc[NT - 1 + 7] = agc(s, p, c.d.z, c.d.w);
            Blk8 o{};
            if (write)
            {
                v2f acc[8];
#pragma unroll
                for (int r = 0; r < 8; r++)
                    acc[r] = v2f{0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < NT; j++)
                {
                    // two taps share one 64-bit register pair; the packed multiply takes either half for both of its lanes (op_sel)
                    const v2f tp{sd_to_vgpr(p.taps[j & ~1]), sd_to_vgpr(p.taps[j | 1])};
                    const v2f tt = (j & 1) ? __builtin_shufflevector(tp, tp, 1, 1) : __builtin_shufflevector(tp, tp, 0, 0);
                    if constexpr (FAST)
                    {
#pragma unroll
                        for (int r = 0; r < 8; r++)
                            acc[r] = __builtin_elementwise_fma(loc[r + j], tt, acc[r]); // v_pk_fma_f32
                    }
                    else
                    {
                        v2f prod[8]; // the eight products first, then the eight sums: independent neighbours for the VALU pipeline
#pragma unroll
                        for (int r = 0; r < 8; r++)
                            prod[r] = loc[r + j] * tt;
#pragma unroll
                        for (int r = 0; r < 8; r++)
                            acc[r] = acc[r] + prod[r];
                    }
                }
                o = Blk8{make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y), make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y),
                         make_float4(acc[4].x, acc[4].y, acc[5].x, acc[5].y), make_float4(acc[6].x, acc[6].y, acc[7].x, acc[7].y)};
            }
#pragma unroll
            for (int i = 0; i < NT - 1; i++)
            {
                s.w[2 * i] = loc[i + 8].x;
                s.w[2 * i + 1] = loc[i + 8].y;
            }
            s.lag[3] = s.lag[2];
            s.lag[2] = s.lag[1];
            s.lag[1] = s.lag[0];
            s.lag[0] = g_in; // gain at the end of the previous block; lag[3] = the gain 32 samples in front of the end of this one
            return o;
        }
        // one sample at a time (ragged tail of a call): same arithmetic, the window moves by one
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v)
        {
            const v2f a = agc(s, p, v.re, v.im);
            v2f acc{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NT - 1; j++)
            {
                const v2f prod = v2f{s.w[2 * j], s.w[2 * j + 1]} * v2f{p.taps[j], p.taps[j]};
                acc = acc + prod;
            }
            {
                const v2f prod = a * v2f{p.taps[NT - 1], p.taps[NT - 1]};
                acc = acc + prod;
            }
#pragma unroll
            for (int i = 0; i < NT - 2; i++)
            {
                s.w[2 * i] = s.w[2 * i + 2];
                s.w[2 * i + 1] = s.w[2 * i + 3];
            }
            s.w[2 * (NT - 2)] = a.x;
            s.w[2 * (NT - 2) + 1] = a.y;
            return cf32{acc.x, acc.y};
        }
    };
    using AgcFirStage = AgcFirStageT<false>;

    // ORDER (2 / 4 / 8) is a template argument: the detector's form is fixed at compile time, so the per-sample loop carries no
    // test of it and none of the other detectors' code.
    template <int ORDER, int D = 4, bool FAST = false, bool BCLIP = false>
    struct CostasStage
    {
        using P = CostasParams;
        using S = CostasState;
        static constexpr int DEPTH = D; // blocks per load group (2: one 128-byte line); two groups in flight
        __device__ static __forceinline__ S init(const P &p, int) { return S{0.0f, p.init_freq}; }
        // early exit of a re-run lane (CKPT): same frame (the engine aligns a re-run with the speculative run's frame), phase
        // compared modulo the loop's own 2 pi wrap, the Costas certificate's windows
        __device__ static __forceinline__ bool close(const S &a, const S &b, float tol_phase, float tol_freq)
        {
            const float twopi = 6.28318530717958647692f;
            float d = a.phase - b.phase;
            d -= twopi * rintf(d / twopi);
            return fabsf(d) < tol_phase && fabsf(a.freq - b.freq) < tol_freq;
        }
        // Start phase of a warm-up from a feed-forward M-th power estimate over its first est_len samples, so that the loop
        // starts next to one of its `order` stable points instead of anywhere in between: a restart that lands near the
        // unstable point half way hangs there for many time constants (the cause of nearly all Costas re-runs at pll_bw
        // 0.002-0.003). Speculation only -- the boundary certificate decides.
        __device__ static __forceinline__ void prewarm(S &s, const P &p, const cf32 *x, long long i0)
        {
            if (p.est_len <= 0 || p.order > 4)
                return;
            const int M = p.order;
            float rc, rs; // exp(-j*M*freq*n), advanced by recurrence
            __sincosf(-(float)M * s.freq, &rs, &rc);
            float cr = 1.0f, ci = 0.0f, ar = 0.0f, ai = 0.0f;
            for (int n = 0; n < p.est_len; n++)
            {
                const cf32 v = x[i0 + n];
                float zr = v.re * v.re - v.im * v.im, zi = 2.0f * v.re * v.im; // x^2
                if (M == 4)
                {
                    const float tr = zr * zr - zi * zi, ti = 2.0f * zr * zi;
                    zr = tr;
                    zi = ti;
                }
                ar += zr * cr - zi * ci;
                ai += zr * ci + zi * cr;
                const float nr = cr * rc - ci * rs, ni = cr * rs + ci * rc;
                cr = nr;
                ci = ni;
            }
            // BPSK symbols sit on the real axis (x^2 -> +1), QPSK symbols on the diagonals (x^4 -> -1)
            const float ang = (M == 4) ? atan2f(-ai, -ar) : atan2f(ai, ar);
            s.phase = ang / (float)M;
        }
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v)
        {
            // CostasLoopBlock::work, costas_loop.cpp:23-65
            float cs, sn; // cosf(-phase), sinf(-phase): two libm calls in the reference (costas_loop.cpp:26), one fused evaluation here
            if constexpr (FAST)
                sd_sincosf_fast(-s.phase, sn, cs);
            else
                sd_sincosf(-s.phase, sn, cs);
            const float tr = (v.re * cs) - (v.im * sn);
            const float ti = (v.im * cs) + (v.re * sn);
            float error;
            if constexpr (ORDER == 2)
                error = tr * ti;
            else if constexpr (ORDER == 4)
                error = (tr > 0.0f ? 1.0f : -1.0f) * ti - (ti > 0.0f ? 1.0f : -1.0f) * tr;
            else
            {
                const float K = sqrtf(2.0f) - 1.0f; // (sqrtf(2.0) - 1)
                const float st = tr > 0.0f ? 1.0f : -1.0f, su = ti > 0.0f ? 1.0f : -1.0f;
                const float ea = st * ti - su * tr * K, eb = st * ti * K - su * tr;
                error = fabsf(tr) >= fabsf(ti) ? ea : eb;
            }
            if constexpr (BCLIP)
                error = error < -1.0f ? -1.0f : (error > 1.0f ? 1.0f : error); // dsp::branched_clip(error, 1.0), block.cpp:7-15 (ndsp CostasBlock)
            else
                error = 0.5f * (fabsf(error + 1.0f) - fabsf(error - 1.0f)); // branchless_clip(error, 1.0), block.cpp:5
            s.freq = s.freq + p.beta * error;
            s.phase = s.phase + (s.freq + p.alpha * error);
            // while (phase > 2 pi) phase -= 2 pi; while (phase < -2 pi) phase += 2 pi; -- the float compared with the double constant,
            // the step taken in double and rounded back (costas_loop.cpp:55-58). The phase was inside [-2 pi, 2 pi] before this step
            // and one step moves it by at most fmax + beta + alpha, which the engine requires to be under 2 pi (DemodEngine ctor): each
            // loop runs at most once, so they are two selects. "float > 2 pi (double)" is "float >= (float)(2 pi)": the float next
            // above the double constant is its own rounding, 6.2831854820251465.
            const double twopi = 2 * 3.14159265358979323846;
            {
                // ONE double addition of -2 pi, 0 or +2 pi: adding 0.0 and rounding back returns the float itself
                const float twopi_f = 6.28318548202514648f;
                const double step = s.phase >= twopi_f ? -twopi : (s.phase <= -twopi_f ? twopi : 0.0);
                s.phase = (float)((double)s.phase + step);
            }
            // if (freq > fmax) freq = fmax; if (freq < fmin) freq = fmin; (fmin <= fmax: v_min / v_max)
            s.freq = __builtin_fmaxf(__builtin_fminf(s.freq, p.fmax), p.fmin);
            return cf32{tr, ti};
        }
    };

    // ---- carrier-tracking PLL of psk_demod's has_carrier mode (PLLCarrierTrackingBlock::work, pll_carrier_tracking.cpp:23-66) ----------
    // Phase detector = the table-driven arctangent of GNU Radio (fast_trig.cpp:60-153: y/x folded into 0..1, 255 intervals, linear
    // interpolation, the octant put back), VCO = two polynomials evaluated in double (fast_trig.cpp:157-180). Written here as
    // selects over the octant instead of the reference's nest of branches; every float / double rounding sits where the reference's
    // declarations put it.
    __device__ __forceinline__ float sd_fast_atan2f(float y, float x, const float *__restrict__ tab)
    {
        const float ya = fabsf(y), xa = fabsf(x);
        if (!(ya > 0.0f || xa > 0.0f))
            return 0.0f;
        const bool flat = ya < xa;              // ratio taken the way that keeps it <= 1 (ties: x/y, like the reference)
        const float z = flat ? ya / xa : xa / ya;
        float base = z;                          // below the table's resolution the angle is its own tangent
        if (!((double)z < 0.003921569))
        {
            const float al = z * 255.0f;
            const int idx = ((int)al) & 0xff;
            const float lo = tab[idx], hi = tab[idx + 1];
            base = lo + (hi - lo) * (al - (float)idx);
        }
        const float pi = (float)3.14159265358979323846, half = (float)1.57079632679489661923;
        if (xa > ya) // within 45 degrees of the x axis
            return x >= 0.0f ? (y >= 0.0f ? base : -base) : (y >= 0.0f ? pi - base : base - pi);
        // within 45 degrees of the y axis
        return y >= 0.0f ? (x >= 0.0f ? half - base : half + base) : (x >= 0.0f ? -half + base : -half - base);
    }
    __device__ __forceinline__ float sd_fast_cos(float x)
    {
        const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
        return (float)((-2.7236370439787708e-7 * x2 + 2.4799852696610628e-5) * x8 + (-1.3888885054799695e-3 * x2 + 4.1666666636943683e-2) * x4 +
                       (-4.9999999999963024e-1 * x2 + 1.0000000000000000e+0));
    }
    __device__ __forceinline__ float sd_fast_sin(float x)
    {
        const float x2 = x * x, x4 = x2 * x2;
        return (float)(((2.7181216275479732e-6 * x2 - 1.9839312269456257e-4) * x4 + (8.3333293048425631e-3 * x2 - 1.6666666640797048e-1)) * x2 * x + x);
    }
    // float value wrapped into [-pi, pi] the reference's way: compared and stepped in double, stored back as float each step
    __device__ __forceinline__ float sd_wrap_pi(float v)
    {
        const double pi = 3.14159265358979323846, twopi = 2 * 3.14159265358979323846;
        while ((double)v < -pi)
            v = (float)((double)v + twopi);
        while ((double)v > pi)
            v = (float)((double)v - twopi);
        return v;
    }
    struct PllStage
    {
        using P = PllParams;
        using S = CostasState;
        static constexpr int DEPTH = 2;
        __device__ static __forceinline__ S init(const P &p, int) { return S{0.0f, p.init_freq}; }
        __device__ static __forceinline__ bool close(const S &a, const S &b, float tol_phase, float tol_freq)
        {
            const float twopi = 6.28318530717958647692f;
            float d = a.phase - b.phase;
            d -= twopi * rintf(d / twopi);
            return fabsf(d) < tol_phase && fabsf(a.freq - b.freq) < tol_freq;
        }
        // a warm-up starts ON the carrier: the loop's phase detector is arg(x) - phase, so arg of the first sample is the start
        // phase with zero error (speculation only -- the boundary certificate decides)
        __device__ static __forceinline__ void prewarm(S &s, const P &p, const cf32 *x, long long i0)
        {
            const cf32 v = x[i0];
            s.phase = sd_fast_atan2f(v.im, v.re, p.atan_tab);
        }
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v)
        {
            const float vr = sd_fast_cos(s.phase), vi = -sd_fast_sin(s.phase);
            const cf32 o{(v.re * vr) - (v.im * vi), (v.im * vr) + (v.re * vi)};
            const float pe = sd_wrap_pi(sd_fast_atan2f(v.im, v.re, p.atan_tab) - s.phase);
            s.freq = s.freq + p.beta * pe;
            if (s.freq > p.fmax)
                s.freq = p.fmax;
            else if (s.freq < p.fmin)
                s.freq = p.fmin;
            s.phase = sd_wrap_pi(s.phase + s.freq + p.alpha * pe);
            return o;
        }
    };

    // ---- ndsp::CostasFastBlock::process_order<ORDER> (dsp/pll/costas_fast.cpp:15-89): one lane, the reference's float operations in its order (products and
    // sums rounded one by one: the library is built with -ffp-contract=off). The frequency limiter hands the rate phasor the OPPOSITE limit's value
    // (costas_fast.cpp:81-84: freq = max goes with freq_limit_min_cpx) -- kept, it is what the block does.
    __device__ __forceinline__ float sd_fast_invsqrt(float x)
    { // fast_math.h:9-18
        const float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(x) >> 1));
        return y * (1.5f - ((0.5f * x) * y) * y);
    }
    template <int ORDER>
    struct CostasFastStage
    {
        using P = CostasFastParams;
        using S = CostasFastState;
        static constexpr int DEPTH = 4;
        // a warm-up lane (chunk-parallel schedule; speculation only -- the hand-off certificate decides): pha = exp(-j phase), fre = exp(-j freq)
        __device__ static __forceinline__ S init(const P &p, int)
        {
            float sn, cs;
            __sincosf(p.init_freq, &sn, &cs);
            return S{p.init_freq, 1.0f, 0.0f, cs, -sn, 0u, 3.0e38f};
        }
        __device__ static __forceinline__ bool close(const S &, const S &, float, float) { return false; }
        __device__ static __forceinline__ void prewarm(S &s, const P &p, const cf32 *x, long long i0)
        {
            s.ctr = (unsigned)(((long long)p.ctr_base + i0) % 65);
            if (p.est_len <= 0 || p.order > 4)
                return;
            CostasState c{0.0f, s.freq};
            CostasParams cp{};
            cp.order = p.order;
            cp.est_len = p.est_len;
            CostasStage<ORDER == 8 ? 4 : ORDER>::prewarm(c, cp, x, i0); // the M-th power estimate of the plain loop's lanes
            float sn, cs;
            __sincosf(c.phase, &sn, &cs);
            s.pha_re = cs;
            s.pha_im = -sn;
        }
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v)
        {
            const float tr = (v.re * s.pha_re) - (v.im * s.pha_im);
            const float ti = (v.re * s.pha_im) + (v.im * s.pha_re);
            float error;
            if constexpr (ORDER == 2)
                error = tr * ti;
            else if constexpr (ORDER == 4)
                error = ((tr > 0.0f ? 1.0f : -1.0f) * ti) - ((ti > 0.0f ? 1.0f : -1.0f) * tr);
            else
            {
                const float K = 0.41421356f;
                const float a = tr > 0.0f ? 1.0f : -1.0f, b = ti > 0.0f ? 1.0f : -1.0f;
                if (fabsf(tr) >= fabsf(ti))
                    error = (a * ti) - ((b * tr) * K);
                else
                    error = ((a * ti) * K) - (b * tr);
            }
            error = error < -1.0f ? -1.0f : (error > 1.0f ? 1.0f : error); // dsp::branched_clip(error, 1.0f)
            s.freq = s.freq + (p.beta * error);
            const float df = p.beta * error;
            const float nfr = s.fre_re + (df * s.fre_im);
            const float nfi = s.fre_im - (df * s.fre_re);
            s.fre_re = nfr;
            s.fre_im = nfi;
            const float pa = p.alpha * error;
            const float ore = s.pha_re + (pa * s.pha_im);
            const float oim = s.pha_im - (pa * s.pha_re);
            const float npr = (ore * s.fre_re) - (oim * s.fre_im);
            const float npi = (ore * s.fre_im) + (oim * s.fre_re);
            s.pha_re = npr;
            s.pha_im = npi;
            if (s.ctr++ >= 64u)
            {
                s.ctr = 0;
                float inv = sd_fast_invsqrt((s.pha_re * s.pha_re) + (s.pha_im * s.pha_im));
                s.pha_re *= inv;
                s.pha_im *= inv;
                inv = sd_fast_invsqrt((s.fre_re * s.fre_re) + (s.fre_im * s.fre_im));
                s.fre_re *= inv;
                s.fre_im *= inv;
                s.margin = fminf(s.margin, fminf(p.fmax - s.freq, s.freq - p.fmin));
                if (s.freq > p.fmax)
                {
                    s.freq = p.fmax;
                    s.fre_re = p.lim_min_re;
                    s.fre_im = p.lim_min_im;
                }
                if (s.freq < p.fmin)
                {
                    s.freq = p.fmin;
                    s.fre_re = p.lim_max_re;
                    s.fre_im = p.lim_max_im;
                }
            }
            return cf32{tr, ti};
        }
    };

    struct DcStage
    {
        using P = DcParams;
        using S = DcState;
        static constexpr int DEPTH = 4;
        __device__ static __forceinline__ S init(const P &p, int k) { return p.starts[k]; } // the scan's value at this chunk's start
        __device__ static __forceinline__ bool close(const S &a, const S &b, float tol, float)
        {
            const float m = fmaxf(fabsf(b.acc_re), fabsf(b.acc_im));
            return fabsf(a.acc_re - b.acc_re) <= tol * m && fabsf(a.acc_im - b.acc_im) <= tol * m;
        }
        __device__ static __forceinline__ void prewarm(S &, const P &, const cf32 *, long long) {}
        __device__ static __forceinline__ cf32 step(S &s, const P &, const cf32 v)
        { // CorrectIQBlock<complex_t>::work, correct_iq.cpp:27-31 (alpha = 1e-4f, beta = 1.0f - alpha)
            const float alpha = 0.0001f, beta = 1.0f - 0.0001f;
            s.acc_re = s.acc_re * beta + v.re * alpha;
            s.acc_im = s.acc_im * beta + v.im * alpha;
            return cf32{v.re - s.acc_re, v.im - s.acc_im};
        }
    };

    // Run one lane over [i0, i1) of its chunk. The lane's samples are contiguous in memory, so they are moved in 64-byte
    // blocks (8 samples = 4 x float4) through a register queue Stage::DEPTH blocks deep: the loads of block j+DEPTH are
    // issued as soon as block j has been consumed, so the ~1.3 us latency of these chunk-strided HBM reads stays off the
    // dependent recurrence (one lane per chunk means few waves per SIMD: nothing else would hide it).
    // x + 8*m must be 16-byte aligned (stage buffers are).
    __device__ __forceinline__ Blk8 blk_load(const cf32 *x, long long i)
    {
        const float4 *xp = reinterpret_cast<const float4 *>(x + i);
        return Blk8{xp[0], xp[1], xp[2], xp[3]};
    }
    // run one block, results returned (stored later, a whole load group's worth at a time: 128/256 contiguous bytes per lane)
    // a stage that defines block() works on a whole 8-sample block at a time (its state holds more than a sample's worth of
    // history, e.g. a filter window); `write` tells it whether the block's output is wanted at all (warm-up blocks: not)
    template <class Stage, class = void>
    struct stage_is_blockwise : std::false_type
    {
    };
    template <class Stage>
    struct stage_is_blockwise<Stage, std::void_t<decltype(&Stage::block)>> : std::true_type
    {
    };
    template <class Stage>
    __device__ __forceinline__ Blk8 blk_step(typename Stage::S &s, const typename Stage::P &p, const Blk8 &c, bool write)
    {
        if constexpr (stage_is_blockwise<Stage>::value)
            return Stage::block(s, p, c, write);
        else
        {
            const cf32 a0 = Stage::step(s, p, cf32{c.a.x, c.a.y});
            const cf32 a1 = Stage::step(s, p, cf32{c.a.z, c.a.w});
            const cf32 a2 = Stage::step(s, p, cf32{c.b.x, c.b.y});
            const cf32 a3 = Stage::step(s, p, cf32{c.b.z, c.b.w});
            const cf32 a4 = Stage::step(s, p, cf32{c.c.x, c.c.y});
            const cf32 a5 = Stage::step(s, p, cf32{c.c.z, c.c.w});
            const cf32 a6 = Stage::step(s, p, cf32{c.d.x, c.d.y});
            const cf32 a7 = Stage::step(s, p, cf32{c.d.z, c.d.w});
            return Blk8{make_float4(a0.re, a0.im, a1.re, a1.im), make_float4(a2.re, a2.im, a3.re, a3.im), make_float4(a4.re, a4.im, a5.re, a5.im),
                        make_float4(a6.re, a6.im, a7.re, a7.im)};
        }
    }
    template <class Stage, int D>
    __device__ __forceinline__ void group_run(typename Stage::S &s, const typename Stage::P &p, const Blk8 (&q)[D], cf32 *y, long long i, bool write)
    {
        Blk8 o[D];
#pragma unroll
        for (int d = 0; d < D; d++)
            o[d] = blk_step<Stage>(s, p, q[d], write);
        if (write)
        {
            float4 *yp = reinterpret_cast<float4 *>(y + i);
#pragma unroll
            for (int d = 0; d < D; d++)
            {
                yp[4 * d + 0] = o[d].a;
                yp[4 * d + 1] = o[d].b;
                yp[4 * d + 2] = o[d].c;
                yp[4 * d + 3] = o[d].d;
            }
        }
    }
    // Groups of Stage::DEPTH blocks, double buffered: all loads of group j+1 (DEPTH * 64 contiguous bytes of this lane's stream,
    // whole 128-byte lines when DEPTH is even and the range starts on a 16-sample boundary) are issued together before group
    // j is consumed, instead of one 64-byte block at a time.
    // HOOK: hook(i) is called whenever the lane has just finished the samples in front of i and (i - hb) is a multiple of hstep
    // (a power of two, a multiple of the group size; i0 and hb are multiples of 64): the checkpoints of k_chunks. It returns
    // true to stop the lane there. The test rides on the group loop -- a mask and a compare per 16 / 32 samples -- so that the
    // main launch keeps its double-buffered stream instead of restarting it for every piece.
    struct NoHook
    {
        __device__ __forceinline__ bool operator()(long long) const { return false; }
    };
    template <class Stage, bool HOOK = false, class Hook = NoHook>
    __device__ __forceinline__ void run_range(typename Stage::S &s, const typename Stage::P &p, const cf32 *x, cf32 *y, long long i0, long long i1, bool write,
                                              long long hb = 0, int hstep = 0, Hook hook = Hook())
    {
        constexpr int D = Stage::DEPTH;
        long long i = i0;
        for (; i < i1 && (i & 7); i++)
        {
            const cf32 o = Stage::step(s, p, x[i]);
            if (write)
                y[i] = o;
        }
        if (i + 8 * D <= i1)
        {
            Blk8 qa[D], qb[D];
#pragma unroll
            for (int d = 0; d < D; d++)
                qa[d] = blk_load(x, i + 8 * d);
            for (;;)
            {
                const bool more_b = i + 16 * D <= i1;
                if (more_b)
                {
#pragma unroll
                    for (int d = 0; d < D; d++)
                        qb[d] = blk_load(x, i + 8 * (D + d));
                }
                group_run<Stage, D>(s, p, qa, y, i, write);
                i += 8 * D;
                if constexpr (HOOK)
                    if ((((i - hb) & (long long)(hstep - 1)) == 0) && i < i1 && hook(i))
                        return;
                if (!more_b)
                    break;
                const bool more_a = i + 16 * D <= i1;
                if (more_a)
                {
#pragma unroll
                    for (int d = 0; d < D; d++)
                        qa[d] = blk_load(x, i + 8 * (D + d));
                }
                group_run<Stage, D>(s, p, qb, y, i, write);
                i += 8 * D;
                if constexpr (HOOK)
                    if ((((i - hb) & (long long)(hstep - 1)) == 0) && i < i1 && hook(i))
                        return;
                if (!more_a)
                    break;
            }
        }
        for (; i < i1; i++)
        {
            const cf32 o = Stage::step(s, p, x[i]);
            if (write)
                y[i] = o;
        }
    }

    // Variant for stages whose block is large and register-hungry (the fused AGC + filter + Costas block: ~1600 instructions, a 38-sample
    // window and 31 taps in registers). run_range keeps two statically named register queues and with them two copies of a four-block
    // group -- ~100 KB of loop, more than the 64 KB instruction cache two CUs share, and 128 registers of queue. Here: ONE group body and
    // ONE four-block queue that is refilled in halves -- slots 0,1 (the next group's first 128 bytes) as soon as block 1 has been consumed,
    // slots 2,3 after block 3 -- so a load has two to three blocks of compute (~3 us) to land; every block's output is stored as soon as
    // it exists. Output is stored from sample index `write_from` on (a whole number of groups away from i0): warm-up and chunk are one
    // loop. hook: as run_range's.
    template <class Stage>
    __device__ __forceinline__ void blk_store(cf32 *y, long long i, const Blk8 &o)
    {
        float4 *yp = reinterpret_cast<float4 *>(y + i);
        yp[0] = o.a;
        yp[1] = o.b;
        yp[2] = o.c;
        yp[3] = o.d;
    }
    // ---- coalesced access to the lanes' streams (round 5) ---------------------------------------------------------------------------------
    // A lane-per-chunk stage reads and writes 64 far-apart streams per wave: every 16-byte piece a lane moves is a request of its own to
    // a cache line of its own, and all the lane stages ran at the same ~200 G such requests per second whatever their arithmetic (k_afc,
    // k_mm, the stand-alone Costas / AGC stages: 2.9 - 3.2 TB/s of payload) -- the rate of uncoalesced requests was the roof, not VALU issue,
    // not HBM. Here the wave moves the same bytes COOPERATIVELY: the lanes of a wave sit at the same place of their streams (same chunk
    // length, same warm-up), so the 128 bytes every lane needs next are fetched by eight loads in which lanes 8r .. 8r+7 read the eight
    // consecutive 16-byte pieces of stream 8j + r -- eight whole 128-byte lines per instruction instead of 64 sixteen-byte pieces of 64
    // lines -- and are handed to their owners through an LDS transpose (region stride 144 bytes: the owners' 16-byte reads start on
    // disjoint bank groups). Stores go the other way. Same bytes, same values, an eighth of the requests.
    constexpr int COOP_REGION = 144;
    constexpr int COOP_LDS_BYTES = 64 * COOP_REGION;
    struct Coop
    {
        char *lw, *lr, *sw, *sr; // this lane's places in the wave's two LDS transpose buffers: load side write (as loader) / read (as owner), store side write (as owner) / read
        long long stride;        // bytes between the streams of adjacent lanes (= chunk length * sizeof(cf32))
        unsigned voff;           // (lane >> 3) * stride + (lane & 7) * 16: this lane's offset from "stream 8 j of the wave, byte 0 of the burst"
    };
    __device__ __forceinline__ Coop coop_make(char *lds, long long stride)
    {
        const int lane = (int)threadIdx.x & 63;
        Coop co;
        co.lw = lds + (lane >> 3) * COOP_REGION + (lane & 7) * 16;
        co.lr = lds + lane * COOP_REGION;
        co.sw = lds + COOP_LDS_BYTES + lane * COOP_REGION;
        co.sr = lds + COOP_LDS_BYTES + (lane >> 3) * COOP_REGION + (lane & 7) * 16;
        co.stride = stride;
        co.voff = (unsigned)((lane >> 3) * stride + (lane & 7) * 16);
        return co;
    }
    // a value every lane of the wave holds alike, as a scalar (lane 0's copy)
    __device__ __forceinline__ int sd_uniform(int v) { return __builtin_amdgcn_readlane(v, 0); }
    __device__ __forceinline__ long long sd_uniform(long long v)
    {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffll), 0);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), 0);
        return (long long)(((unsigned long long)hi << 32) | lo);
    }
    __device__ __forceinline__ void sd_wave_sync()
    { // one lane's LDS write before another lane's read of it: lockstep on the device (the fence keeps the compiler from moving the accesses), a meeting point on the host twin
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    struct Burst
    {
        float4 c0, c1, c2, c3, c4, c5, c6, c7; // piece (lane & 7) of the streams 8 j + (lane >> 3), j = 0 .. 7 (named members: an array here stayed in scratch memory)
    };
    // the 128 bytes at stream index i0w (LANE 0's index, a scalar; lane l's own index is i0w + l * chunk length) of every stream of the wave
    __device__ __forceinline__ Burst coop_load(const cf32 *x, long long i0w, const Coop &co)
    {
        const char *sb = reinterpret_cast<const char *>(x + i0w);
        const long long st8 = 8 * co.stride;
        Burst b;
        b.c0 = *reinterpret_cast<const float4 *>(sb + co.voff);
        b.c1 = *reinterpret_cast<const float4 *>(sb + st8 + co.voff);
        b.c2 = *reinterpret_cast<const float4 *>(sb + 2 * st8 + co.voff);
        b.c3 = *reinterpret_cast<const float4 *>(sb + 3 * st8 + co.voff);
        b.c4 = *reinterpret_cast<const float4 *>(sb + 4 * st8 + co.voff);
        b.c5 = *reinterpret_cast<const float4 *>(sb + 5 * st8 + co.voff);
        b.c6 = *reinterpret_cast<const float4 *>(sb + 6 * st8 + co.voff);
        b.c7 = *reinterpret_cast<const float4 *>(sb + 7 * st8 + co.voff);
        return b;
    }
    __device__ __forceinline__ void coop_unpack(const Burst &b, const Coop &co, Blk8 &q0, Blk8 &q1)
    {
        __builtin_amdgcn_sched_barrier(0); // (the transposes stay where they are written: hoisted in front of the preceding blocks' arithmetic they kept two more blocks alive)
        *reinterpret_cast<float4 *>(co.lw + 0 * 8 * COOP_REGION) = b.c0;
        *reinterpret_cast<float4 *>(co.lw + 1 * 8 * COOP_REGION) = b.c1;
        *reinterpret_cast<float4 *>(co.lw + 2 * 8 * COOP_REGION) = b.c2;
        *reinterpret_cast<float4 *>(co.lw + 3 * 8 * COOP_REGION) = b.c3;
        *reinterpret_cast<float4 *>(co.lw + 4 * 8 * COOP_REGION) = b.c4;
        *reinterpret_cast<float4 *>(co.lw + 5 * 8 * COOP_REGION) = b.c5;
        *reinterpret_cast<float4 *>(co.lw + 6 * 8 * COOP_REGION) = b.c6;
        *reinterpret_cast<float4 *>(co.lw + 7 * 8 * COOP_REGION) = b.c7;
        sd_wave_sync();
        const float4 *r = reinterpret_cast<const float4 *>(co.lr);
        q0 = Blk8{r[0], r[1], r[2], r[3]};
        q1 = Blk8{r[4], r[5], r[6], r[7]};
        sd_wave_sync();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void coop_store_half(const Blk8 &o, int half, const Coop &co)
    {
        float4 *w = reinterpret_cast<float4 *>(co.sw + half * 64);
        w[0] = o.a;
        w[1] = o.b;
        w[2] = o.c;
        w[3] = o.d;
    }
    // the 128 bytes every lane has put together with two coop_store_half calls go out to stream index i0w (lane 0's, as in coop_load)
    __device__ __forceinline__ void coop_store_flush(cf32 *y, long long i0w, const Coop &co)
    {
        __builtin_amdgcn_sched_barrier(0);
        sd_wave_sync();
        char *sb = reinterpret_cast<char *>(y + i0w);
#pragma unroll
        for (int j = 0; j < 8; j++)
            *reinterpret_cast<float4 *>(sb + (long long)(8 * j) * co.stride + co.voff) = *reinterpret_cast<const float4 *>(co.sr + j * 8 * COOP_REGION);
        sd_wave_sync();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <class Stage, bool COOP = false, class Hook>
    __device__ __forceinline__ void run_range1(typename Stage::S &s, const typename Stage::P &p, const cf32 *x, cf32 *y, long long i0, long long i1,
                                               long long write_from, long long hb, int hstep, Hook hook, const Coop *co = nullptr)
    {
        static_assert(Stage::DEPTH == 4, "four-block groups");
        long long i = i0;
        for (; i < i1 && (i & 7); i++)
        {
            const cf32 o = Stage::step(s, p, x[i]);
            if (i >= write_from)
                y[i] = o;
        }
        if (i + 32 <= i1)
        {
            if constexpr (COOP)
            { // every lane of the wave is here with the same i1 - i, the same write_from - i, and nobody leaves through the hook (the caller's promise):
              // the loop runs on scalars (position relative to the start, lane 0's stream index), the lane's own index only names the hook's position
                const long long lim = 1ll << 30;
                const long long wf = write_from - i, tl = i1 - i;
                const int total = sd_uniform((int)(tl < lim ? tl : lim));
                const int wfrom = sd_uniform((int)(wf < -lim ? -lim : (wf < lim ? wf : lim)));
                const int hoff = sd_uniform((int)((i - hb) & (long long)(hstep - 1)));
                const long long iw = sd_uniform(i);
                const long long ibase = i;
                int r = 0;
                Burst b0 = coop_load(x, iw, *co), b1 = coop_load(x, iw + 16, *co);
                Blk8 q[4];
                for (;;)
                {
                    const bool more = r + 64 <= total;
                    const bool wr = r >= wfrom;
                    coop_unpack(b0, *co, q[0], q[1]);
#pragma unroll
                    for (int d = 0; d < 2; d++)
                    {
                        const Blk8 o = blk_step<Stage>(s, p, q[d], wr);
                        if (wr)
                            coop_store_half(o, d, *co);
                    }
                    if (wr)
                        coop_store_flush(y, iw + r, *co);
                    if (more)
                        b0 = coop_load(x, iw + r + 32, *co);
                    coop_unpack(b1, *co, q[2], q[3]);
#pragma unroll
                    for (int d = 2; d < 4; d++)
                    {
                        const Blk8 o = blk_step<Stage>(s, p, q[d], wr);
                        if (wr)
                            coop_store_half(o, d - 2, *co);
                    }
                    if (wr)
                        coop_store_flush(y, iw + r + 16, *co);
                    if (more)
                        b1 = coop_load(x, iw + r + 48, *co);
                    r += 32;
                    if ((((r + hoff) & (hstep - 1)) == 0) && r < total)
                        (void)hook(ibase + r);
                    if (!more)
                        break;
                }
                i = ibase + r;
            }
            else
            {
            Blk8 q[4];
#pragma unroll
            for (int d = 0; d < 4; d++)
                q[d] = blk_load(x, i + 8 * d);
            for (;;)
            {
                const bool more = i + 64 <= i1;
                const bool wr = i >= write_from;
#pragma unroll
                for (int d = 0; d < 4; d++)
                {
                    const Blk8 o = blk_step<Stage>(s, p, q[d], wr);
                    if (wr)
                        blk_store<Stage>(y, i + 8 * d, o);
                    if ((d & 1) && more)
                    {
                        q[d - 1] = blk_load(x, i + 32 + 8 * (d - 1));
                        q[d] = blk_load(x, i + 32 + 8 * d);
                    }
                }
                i += 32;
                if ((((i - hb) & (long long)(hstep - 1)) == 0) && i < i1 && hook(i))
                    return;
                if (!more)
                    break;
            }
            }
        }
        for (; i < i1; i++)
        {
            const cf32 o = Stage::step(s, p, x[i]);
            if (i >= write_from)
                y[i] = o;
        }
    }

    // CKPT: every ck_len samples of its chunk a lane leaves its state in ck[k][*]. A re-run lane compares itself with the
    // checkpoint at the same sample index and stops as soon as Stage::close() holds: from there on the output and the end state
    // of the earlier run stand, under the rule that accepts a chunk boundary. A lane that does not merge overwrites the
    // checkpoints (they always describe the trajectory whose samples are in y).
    template <class Stage, bool CKPT>
    __global__ __launch_bounds__(64) void k_chunks(const cf32 *x, cf32 *y, ChunkGeom g, typename Stage::P p, const typename Stage::S *start0,
                                                   typename Stage::S *spec, typename Stage::S *endst, const int *redo, int nredo, typename Stage::S *ck,
                                                   int ck_per_chunk, int ck_len, float tol_a, float tol_b, unsigned long long *ck_work)
    {
        const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        int k;
        typename Stage::S s;
        if (redo)
        {
            if (idx >= nredo)
                return;
            k = redo[idx];
            // exact state at the chunk boundary: the SNAPSHOT of endst[k-1] the engine put into spec[k] before this launch
            // (k_spec_from_prev), not endst[k-1] itself -- the predecessor may be re-run in this very launch and overwrite it,
            // and the certificate of this chunk is judged against the snapshot
            s = spec[k];
        }
        else
        {
            k = idx;
            if (k >= g.K)
                return;
            if (k == 0)
                s = *start0;
            else
            {
                s = Stage::init(p, k);
                const long long b = chunk_begin(g, k);
                Stage::prewarm(s, p, x, b - g.W);
                run_range<Stage>(s, p, x, y, b - g.W, b, false);
                spec[k] = s;
            }
        }
        const long long b = chunk_begin(g, k), e = chunk_end(g, k);
        if constexpr (!CKPT)
        {
            run_range<Stage>(s, p, x, y, b, e, true);
            endst[k] = s;
        }
        else
        {
            // checkpoint j = the state after the samples in front of b + (j + 1) * ck_len. The main launch leaves them behind; a
            // re-run lane compares itself with the one at the same position and stops as soon as Stage::close() holds.
            bool merged = false;
            int jmax = 0;
            typename Stage::S *cks = ck + (size_t)k * ck_per_chunk;
            run_range<Stage, true>(s, p, x, y, b, e, true, b, ck_len, [&](long long i) -> bool {
                const int j = (int)((i - b) / ck_len) - 1;
                if (j < 0 || j >= ck_per_chunk)
                    return false;
                jmax = j + 1;
                if (redo && Stage::close(s, cks[j], tol_a, tol_b))
                {
                    merged = true;
                    return true;
                }
                cks[j] = s;
                return false;
            });
            if (!merged)
                endst[k] = s;
            if (redo && ck_work)
            { // statistics of the experiment: re-run lanes, pieces they ran, pieces a full re-run would have run
                atomicAdd(ck_work, 1ull);
                atomicAdd(ck_work + 1, (unsigned long long)(jmax + 1));
                atomicAdd(ck_work + 2, (unsigned long long)((e - b + ck_len - 1) / ck_len));
            }
        }
    }

    // =============================================================================================
    // AGC + RRC filter + Costas loop in one lane (demod_kernels.h: AfcParams). Three views of the same lane state for run_range:
    //   AfcAgcOnly : AGC recurrence, the filter window filled on the way, nothing else (first part of a warm-up)
    //   AfcEst<M>  : AGC + filter, the filtered samples raised to the M-th power and summed against the start frequency
    //                (CostasStage::prewarm's estimate, taken on the fly: the filtered samples exist nowhere in memory)
    //   AfcFull<O> : AGC + filter + Costas
    // Arithmetic per stage = AgcFirStage / CostasStage, operation for operation (exact mode: bit for bit the reference).
    // =============================================================================================
    template <bool FAST>
    struct AfcAgcOnly
    {
        using P = AfcParams;
        using S = AfcState;
        static constexpr int DEPTH = 4;
        __device__ static __forceinline__ Blk8 block(S &s, const P &p, const Blk8 &c, bool) { return AgcFirStageT<FAST>::block(s.af, p.af, c, false); }
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v) { return AgcFirStageT<FAST>::step(s.af, p.af, v); }
    };
    struct AfcEstState
    {
        AfcState s;
        float cr, ci, ar, ai, rc, rs; // exp(-j M f n) by recurrence, the running sum, the recurrence's step
    };
    template <int M, bool FAST>
    struct AfcEst
    {
        using P = AfcParams;
        using S = AfcEstState;
        static constexpr int DEPTH = 4;
        __device__ static __forceinline__ void acc(S &e, float re, float im)
        {
            float zr = re * re - im * im, zi = 2.0f * re * im; // x^2
            if constexpr (M == 4)
            {
                const float tr = zr * zr - zi * zi, ti = 2.0f * zr * zi;
                zr = tr;
                zi = ti;
            }
            e.ar += zr * e.cr - zi * e.ci;
            e.ai += zr * e.ci + zi * e.cr;
            const float nr = e.cr * e.rc - e.ci * e.rs, ni = e.cr * e.rs + e.ci * e.rc;
            e.cr = nr;
            e.ci = ni;
        }
        __device__ static __forceinline__ Blk8 block(S &e, const P &p, const Blk8 &c, bool)
        {
            const Blk8 f = AgcFirStageT<FAST>::block(e.s.af, p.af, c, true);
            acc(e, f.a.x, f.a.y);
            acc(e, f.a.z, f.a.w);
            acc(e, f.b.x, f.b.y);
            acc(e, f.b.z, f.b.w);
            acc(e, f.c.x, f.c.y);
            acc(e, f.c.z, f.c.w);
            acc(e, f.d.x, f.d.y);
            acc(e, f.d.z, f.d.w);
            return f;
        }
        __device__ static __forceinline__ cf32 step(S &e, const P &p, const cf32 v)
        {
            const cf32 f = AgcFirStageT<FAST>::step(e.s.af, p.af, v);
            acc(e, f.re, f.im);
            return f;
        }
    };
    template <int ORDER, bool FAST>
    struct AfcFull
    {
        using P = AfcParams;
        using S = AfcState;
        using Cos = CostasStage<ORDER, 4, FAST>;
        static constexpr int DEPTH = 4;
        __device__ static __forceinline__ Blk8 block(S &s, const P &p, const Blk8 &c, bool)
        {
            const Blk8 f = AgcFirStageT<FAST>::block(s.af, p.af, c, true);
            const cf32 a0 = Cos::step(s.cos, p.cos, cf32{f.a.x, f.a.y});
            const cf32 a1 = Cos::step(s.cos, p.cos, cf32{f.a.z, f.a.w});
            const cf32 a2 = Cos::step(s.cos, p.cos, cf32{f.b.x, f.b.y});
            const cf32 a3 = Cos::step(s.cos, p.cos, cf32{f.b.z, f.b.w});
            const cf32 a4 = Cos::step(s.cos, p.cos, cf32{f.c.x, f.c.y});
            const cf32 a5 = Cos::step(s.cos, p.cos, cf32{f.c.z, f.c.w});
            const cf32 a6 = Cos::step(s.cos, p.cos, cf32{f.d.x, f.d.y});
            const cf32 a7 = Cos::step(s.cos, p.cos, cf32{f.d.z, f.d.w});
            return Blk8{make_float4(a0.re, a0.im, a1.re, a1.im), make_float4(a2.re, a2.im, a3.re, a3.im), make_float4(a4.re, a4.im, a5.re, a5.im),
                        make_float4(a6.re, a6.im, a7.re, a7.im)};
        }
        __device__ static __forceinline__ cf32 step(S &s, const P &p, const cf32 v) { return Cos::step(s.cos, p.cos, AgcFirStageT<FAST>::step(s.af, p.af, v)); }
    };
    __device__ __forceinline__ float sd_wrap_2pi(double ph)
    { // into the loop's own range [-2 pi, 2 pi] (costas_loop.cpp:55-58 keeps it there)
        const double twopi = 2 * 3.14159265358979323846;
        while (ph > twopi)
            ph -= twopi;
        while (ph < -twopi)
            ph += twopi;
        return (float)ph;
    }
    // Checkpoints (ck != nullptr): every ck_len samples of its chunk a lane leaves {gain, gain 32 samples ago, phase, freq}; a re-run lane
    // (exact start state) stops at the first checkpoint at which its AGC has merged with the earlier run's (the AGC certificate's rule)
    // and its carrier loop is inside the Costas windows in the earlier run's frame: from there on the earlier output and end state stand.
    // Warm-up tail and chunk are ONE loop over AfcFull (stores begin at the chunk start, where the state is also left in spec[k]).
    template <int ORDER, bool FAST>
    __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_afc(const cf32 *x, cf32 *y, ChunkGeom g, AfcParams p, const AfcState *start0, AfcState *spec, AfcState *endst,
                                                const int *redo, int nredo, AfcCkpt *ck, int ck_per_chunk, int ck_len, float tol_phase, float tol_freq, int coop_nb)
    {
        __shared__ __attribute__((aligned(16))) char coop_lds[2 * COOP_LDS_BYTES];
        const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        // Main launch with cooperative access (coop_nb > 0, see Coop): blocks 0 .. coop_nb-1 take the chunks 1 + 64 blk + lane -- all of them ordinary chunks, a
        // warm-up of their own in front and the full length (the host's promise), so the 64 lanes of such a wave walk their streams in lockstep one chunk length
        // apart --; chunk 0 (no warm-up: the stream's carried state) and the last few chunks (the very last one may be shorter) follow eight to a block on the
        // per-lane path, so that no wave of the launch runs 64 lanes of it
        const bool coop = !redo && coop_nb > 0 && (int)blockIdx.x < coop_nb;
        const Coop co = coop_make(coop_lds, (long long)g.L * (long long)sizeof(cf32));
        int k;
        AfcState s;
        bool leave_spec = false;
        long long i0;
        if (redo)
        {
            if (idx >= nredo)
                return;
            k = redo[idx];
            s = spec[k]; // the engine put the exact boundary state here before this launch (see k_chunks)
            i0 = chunk_begin(g, k);
        }
        else
        {
            if (coop_nb > 0)
            {
                if (coop)
                    k = 1 + idx;
                else
                {
                    const int j = ((int)blockIdx.x - coop_nb) * 8 + (int)threadIdx.x;
                    if ((int)threadIdx.x >= 8 || j >= g.K - 64 * coop_nb)
                        return;
                    k = j == 0 ? 0 : 64 * coop_nb + j;
                }
            }
            else
            {
                k = idx;
                if (k >= g.K)
                    return;
            }
            i0 = 0;
            s = *start0; // chunk 0, and chunks whose warm-up would reach in front of the call: from the stream's true state
            const long long b = chunk_begin(g, k);
            const long long w0 = b - g.W - p.w_agc;
            if (k > 0)
                leave_spec = true;
            if (k > 0 && w0 > 0)
            {
                const auto nohook = [](long long) { return false; };
                const long long never = 1ll << 62;
                s.af = AgcFirStage::init(p.af, k);
                s.cos = CostasState{0.0f, p.cos.init_freq};
                if (coop)
                    run_range1<AfcAgcOnly<FAST>, true>(s, p, x, y, w0, b - g.W, never, 0, 1 << 30, nohook, &co);
                else
                    run_range1<AfcAgcOnly<FAST>>(s, p, x, y, w0, b - g.W, never, 0, 1 << 30, nohook);
                i0 = b - g.W;
                if (p.cos.est_len > 0 && ORDER <= 4)
                {
                    constexpr int M = ORDER <= 2 ? 2 : 4;
                    AfcEstState e;
                    e.s = s;
                    __sincosf(-(float)M * s.cos.freq, &e.rs, &e.rc);
                    e.cr = 1.0f;
                    e.ci = 0.0f;
                    e.ar = 0.0f;
                    e.ai = 0.0f;
                    if (coop)
                        run_range1<AfcEst<M, FAST>, true>(e, p, x, y, i0, i0 + p.cos.est_len, never, 0, 1 << 30, nohook, &co);
                    else
                        run_range1<AfcEst<M, FAST>>(e, p, x, y, i0, i0 + p.cos.est_len, never, 0, 1 << 30, nohook);
                    s = e.s;
                    // BPSK symbols sit on the real axis (x^2 -> +1), QPSK symbols on the diagonals (x^4 -> -1); the sum's angle is M
                    // times the carrier phase at the first sample of the window, the loop takes over est_len samples later
                    const float ang = (M == 4) ? atan2f(-e.ai, -e.ar) : atan2f(e.ai, e.ar);
                    s.cos.phase = sd_wrap_2pi((double)(ang / (float)M) + (double)s.cos.freq * (double)p.cos.est_len);
                    i0 += p.cos.est_len;
                }
            }
        }
        const long long b = chunk_begin(g, k), e = chunk_end(g, k);
        bool merged = false;
        AfcCkpt *cks = ck ? ck + (size_t)k * ck_per_chunk : nullptr;
        const auto hook = [&](long long i) -> bool {
            if (i < b)
                return false;
            if (i == b)
            {
                if (leave_spec)
                    spec[k] = s;
                return false;
            }
            const int j = (int)((i - b) / ck_len) - 1;
            if (!cks || j >= ck_per_chunk)
                return false;
            if (redo)
            {
                const AfcCkpt o = cks[j];
                if (__float_as_uint(o.gain) == __float_as_uint(s.af.gain) && __float_as_uint(o.lag3) == __float_as_uint(s.af.lag[3]) &&
                    CostasStage<ORDER, 4>::close(s.cos, CostasState{o.phase, o.freq}, tol_phase, tol_freq))
                {
                    merged = true;
                    return true;
                }
            }
            cks[j] = AfcCkpt{s.af.gain, s.af.lag[3], s.cos.phase, s.cos.freq};
            return false;
        };
        if (coop) // (never a re-run launch: the hook stops nobody)
            run_range1<AfcFull<ORDER, FAST>, true>(s, p, x, y, i0, e, b, b, ck_len, hook, &co);
        else
            run_range1<AfcFull<ORDER, FAST>>(s, p, x, y, i0, e, b, b, ck_len, hook);
        if (!merged)
            endst[k] = s;
    }
    void launch_afc(const cf32 *x, cf32 *y, const ChunkGeom &g, const AfcParams &p, const AfcState *start0, AfcState *spec, AfcState *endst, const int *redo,
                    int nredo, hipStream_t st, const AfcCkptCfg &ck, bool fast)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_afc", st);
        std::optional<ProfScope> _pr;
        if (redo)
            _pr.emplace("k_afc (re-run launches, included in k_afc)", st);
        auto go = [&](auto order, auto fm) {
            constexpr int O = decltype(order)::value;
            constexpr bool F = decltype(fm)::value;
            // ck_len is also the spacing at which the lane looks for the chunk start (spec snapshot): always a power of two >= 32
            // cooperative access of the wave's 64 streams (see Coop): every stage range a whole number of 128-byte bursts (chunk length, warm-ups, estimator
            // window: multiples of 16 samples), every chunk from 1 on with its whole warm-up inside the call, and at least one full wave of ordinary chunks
            const bool coop_env = !(getenv("SDHIP_COOP") && atoi(getenv("SDHIP_COOP")) == 0); // (read per launch: tests and tools/ab_demod.py switch it in one process)
            int coop_nb = 0;
            if (coop_env && !redo && g.L % 16 == 0 && g.W % 16 == 0 && p.w_agc % 16 == 0 && p.cos.est_len % 16 == 0 && (long long)g.L > (long long)p.w_agc && g.K >= 66)
                coop_nb = (g.K - 2) / 64; // chunks 1 .. 64 coop_nb; chunk 0 and the rest (the last chunk among them) on the per-lane path
            const int nblk = coop_nb > 0 ? coop_nb + (g.K - 64 * coop_nb + 7) / 8 : (n + 63) / 64;
            if (!redo && coop_nb == 0 && g.K >= 66 && getenv("SDHIP_COOP_REQUIRE")) // tests: the path under test must be the one that runs
                throw HipError("k_afc: cooperative access asked for (SDHIP_COOP_REQUIRE) but the geometry does not allow it");
            if (getenv("SDHIP_DEBUG") && !redo)
                fprintf(stderr, "[sdhip] k_afc: K %d L %d W %d w_agc %d est %d -> %d cooperative blocks of %d\n", g.K, g.L, g.W, p.w_agc, p.cos.est_len, coop_nb, nblk);
            hipLaunchKernelGGL((k_afc<O, F>), dim3(nblk), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, ck.ck, ck.per_chunk,
                               ck.len > 0 ? ck.len : 2048, ck.tol_phase, ck.tol_freq, coop_nb);
        };
        auto by_order = [&](auto fm) {
            if (p.cos.order == 2)
                go(std::integral_constant<int, 2>{}, fm);
            else if (p.cos.order == 4)
                go(std::integral_constant<int, 4>{}, fm);
            else
                go(std::integral_constant<int, 8>{}, fm);
        };
        if (fast)
            by_order(std::true_type{});
        else
            by_order(std::false_type{});
    }

    // B_k = sum over the chunk of beta^(len-1-i) * alpha * x_i, in double: thread t takes the samples i = t (mod 256) -- coalesced --
    // with the weight advanced by beta^-256 from one to the next (the exponent only spans the chunk, a few 10^4)
    __global__ __launch_bounds__(256) void k_dc_partial(const cf32 *x, ChunkGeom g, double *partial)
    {
        __shared__ double sr[256], si[256];
        const int k = (int)blockIdx.x, t = (int)threadIdx.x;
        const long long b = chunk_begin(g, k), e = chunk_end(g, k);
        const double beta = (double)(1.0f - 0.0001f), alpha = (double)0.0001f;
        const double up = pow(beta, -256.0);
        double ar = 0, ai = 0;
        if (b + t < e)
        {
            double w = pow(beta, (double)(e - 1 - (b + t))) * alpha;
            for (long long i = b + t; i < e; i += 256)
            {
                const cf32 v = x[i];
                ar += w * (double)v.re;
                ai += w * (double)v.im;
                w *= up; // i + 256 is 256 steps closer to the chunk end
            }
        }
        sr[t] = ar;
        si[t] = ai;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1)
        {
            if (t < s2)
            {
                sr[t] += sr[t + s2];
                si[t] += si[t + s2];
            }
            __syncthreads();
        }
        if (t == 0)
        {
            partial[2 * k] = sr[0];
            partial[2 * k + 1] = si[0];
        }
    }
    // The composed gain map of every chunk (demod_kernels.h: launch_agc_partial). One block per chunk; it walks the chunk in tiles of 2048 samples: the tile's
    // magnitudes |x| are formed from coalesced float4 loads (two samples each) and parked in LDS, thread t composes the maps of samples 8 t .. 8 t + 7 of the tile
    // in order, the 256 per-thread maps are composed by an ordered tree in LDS, thread 0 appends the tile's map to the chunk's.
    // (|x| in float like the reference's own sqrt of the float products; the map coefficients and their composition in double.)
    struct AgcMap
    {
        double a, b, c;
    };
    __device__ __forceinline__ AgcMap agc_map_then(const AgcMap &f1, const AgcMap &f2) // first f1, then f2
    {
        return AgcMap{f2.a * f1.a, f2.a * f1.b + f2.b, fmin(f2.a * f1.c + f2.b, f2.c)};
    }
    constexpr int AGCP_TILE = 2048;
    __global__ __launch_bounds__(256) void k_agc_partial(const cf32 *x, ChunkGeom g, AgcParams p, double *partial)
    {
        __shared__ float mag[AGCP_TILE];
        __shared__ double sa[256], sb[256], sc[256];
        __shared__ int bad;
        const int k = (int)blockIdx.x, t = (int)threadIdx.x;
        const long long b0 = chunk_begin(g, k), e = chunk_end(g, k);
        const double rate = (double)p.rate, rr = (double)p.rate * (double)p.reference;
        const double cmax = p.max_gain > 0.0f ? (double)p.max_gain : __builtin_inf();
        if (t == 0)
            bad = 0;
        const bool aligned = (reinterpret_cast<uintptr_t>(x + b0) & 15) == 0;
        AgcMap acc{1.0, 0.0, __builtin_inf()};
        for (long long tb = b0; tb < e; tb += AGCP_TILE)
        {
            for (int j = 0; j < AGCP_TILE / 512; j++)
            {
                const int s2 = 2 * (t + 256 * j); // sample of the tile this thread's float4 starts at
                const long long i = tb + s2;
                float m0 = 0.0f, m1 = 0.0f;
                if (aligned && i + 2 <= e)
                {
                    const float4 q = *reinterpret_cast<const float4 *>(x + i);
                    m0 = sqrtf(q.x * q.x + q.y * q.y);
                    m1 = sqrtf(q.z * q.z + q.w * q.w);
                }
                else
                {
                    if (i < e)
                        m0 = sqrtf(x[i].re * x[i].re + x[i].im * x[i].im);
                    if (i + 1 < e)
                        m1 = sqrtf(x[i + 1].re * x[i + 1].re + x[i + 1].im * x[i + 1].im);
                }
                mag[s2] = m0;
                mag[s2 + 1] = m1;
            }
            __syncthreads();
            AgcMap m{1.0, 0.0, __builtin_inf()};
            const long long i0 = tb + 8 * t;
            int neg = 0;
            if (i0 < e)
            {
                const int cnt = (int)(e - i0 < 8 ? e - i0 : 8);
                for (int j = 0; j < cnt; j++)
                {
                    const double a = 1.0 - rate * (double)mag[8 * t + j];
                    neg |= (a < 0.0 || !(a == a)) ? 1 : 0;
                    m = agc_map_then(m, AgcMap{a, rr, cmax});
                }
            }
            if (neg)
                atomicOr(&bad, 1);
            sa[t] = m.a, sb[t] = m.b, sc[t] = m.c;
            __syncthreads();
            for (int d = 1; d < 256; d <<= 1)
            { // ordered tree: slot t (t a multiple of 2 d) <- slot t, then slot t + d
                if ((t & (2 * d - 1)) == 0)
                {
                    const AgcMap r = agc_map_then(AgcMap{sa[t], sb[t], sc[t]}, AgcMap{sa[t + d], sb[t + d], sc[t + d]});
                    sa[t] = r.a, sb[t] = r.b, sc[t] = r.c;
                }
                __syncthreads();
            }
            if (t == 0)
                acc = agc_map_then(acc, AgcMap{sa[0], sb[0], sc[0]});
            __syncthreads();
        }
        if (t == 0)
        {
            partial[4 * k] = acc.a;
            partial[4 * k + 1] = acc.b;
            partial[4 * k + 2] = acc.c;
            partial[4 * k + 3] = bad ? 0.0 : 1.0;
        }
    }
    void launch_agc_partial(const cf32 *x, const ChunkGeom &g, const AgcParams &p, double *partial, hipStream_t st)
    {
        ProfScope _ps("k_agc_partial", st);
        hipLaunchKernelGGL(k_agc_partial, dim3(g.K), dim3(256), 0, st, x, g, p, partial);
    }
    void launch_dc_partial(const cf32 *x, const ChunkGeom &g, double *partial, hipStream_t st)
    {
        ProfScope _ps("k_dc_partial", st);
        hipLaunchKernelGGL(k_dc_partial, dim3(g.K), dim3(256), 0, st, x, g, partial);
    }
    void launch_dcblock(const cf32 *x, cf32 *y, const ChunkGeom &g, const DcParams &p, const DcState *start0, DcState *spec, DcState *endst, const int *redo,
                        int nredo, hipStream_t st)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<DcStage>", st);
        hipLaunchKernelGGL((k_chunks<DcStage, false>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (DcState *)nullptr, 0, 0,
                           0.0f, 0.0f, (unsigned long long *)nullptr);
    }
    void launch_agc(const cf32 *x, cf32 *y, const ChunkGeom &g, const AgcParams &p, const AgcState *start0, AgcState *spec, AgcState *endst, const int *redo,
                    int nredo, hipStream_t st, const ChunkCkpt &ck)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<AgcStage>", st);
        static const int depth = [] {
            const char *e = getenv("SDHIP_AGC_DEPTH"); // experiment: bytes per lane per load group = 64 * depth
            return e ? atoi(e) : 4;
        }();
        auto go = [&](auto stage) {
            using St = decltype(stage);
            if (ck.ck)
                hipLaunchKernelGGL((k_chunks<St, true>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (AgcState *)ck.ck,
                                   ck.per_chunk, ck.len, ck.tol_a, ck.tol_b, ck.work);
            else
                hipLaunchKernelGGL((k_chunks<St, false>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (AgcState *)nullptr,
                                   0, 0, 0.0f, 0.0f, (unsigned long long *)nullptr);
        };
        if (p.fast) // chunk-parallel mode's arithmetic (hardware square root, one fma): the serial chain per sample is what a slow loop's
            (getenv("SDHIP_AGC_DEPTH") && depth != 8) ? go(AgcStageT<4, true>{}) : go(AgcStageT<8, true>{}); // (measured: 28.7 -> 25.0 ms with 8 blocks per group) long warm-up costs (the ndsp block's rate 1e-4: 6e5 sequential steps per lane)
        else if (depth == 8)
            go(AgcStageT<8>{});
        else if (depth == 2)
            go(AgcStageT<2>{});
        else
            go(AgcStageT<4>{});
    }
    void launch_agc_fir(const cf32 *x, cf32 *y, const ChunkGeom &g, const AgcFirParams &p, const AgcFirState *start0, AgcFirState *spec, AgcFirState *endst,
                        const int *redo, int nredo, hipStream_t st)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<AgcFirStage>", st);
        hipLaunchKernelGGL((k_chunks<AgcFirStage, false>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (AgcFirState *)nullptr,
                           0, 0, 0.0f, 0.0f, (unsigned long long *)nullptr);
    }
    void launch_costas(const cf32 *x, cf32 *y, const ChunkGeom &g, const CostasParams &p, const CostasState *start0, CostasState *spec, CostasState *endst,
                       const int *redo, int nredo, hipStream_t st, const ChunkCkpt &ck)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<CostasStage>", st);
        auto go = [&](auto stage) {
            using St = decltype(stage);
            if (ck.ck)
                hipLaunchKernelGGL((k_chunks<St, true>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (CostasState *)ck.ck,
                                   ck.per_chunk, ck.len, ck.tol_a, ck.tol_b, ck.work);
            else
                hipLaunchKernelGGL((k_chunks<St, false>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo,
                                   (CostasState *)nullptr, 0, 0, 0.0f, 0.0f, (unsigned long long *)nullptr);
        };
        static const int depth = [] {
            const char *e = getenv("SDHIP_COSTAS_DEPTH"); // experiment: 64-byte blocks per load group (2, 4, 8)
            return e ? atoi(e) : 4;
        }();
        if (p.clip_branched)
        { // the ndsp block's clip (one load-group depth: this is the symbol-rate loop of the hier block)
            if (p.order == 2)
                go(CostasStage<2, 4, false, true>{});
            else if (p.order == 4)
                go(CostasStage<4, 4, false, true>{});
            else
                go(CostasStage<8, 4, false, true>{});
        }
        else if (p.order == 2)
            depth == 2 ? go(CostasStage<2, 2>{}) : go(CostasStage<2, 4>{});
        else if (p.order == 4 && depth == 2)
            go(CostasStage<4, 2>{});
        else if (p.order == 4 && depth == 8)
            go(CostasStage<4, 8>{});
        else if (p.order == 4)
            go(CostasStage<4, 4>{});
        else
            depth == 2 ? go(CostasStage<8, 2>{}) : go(CostasStage<8, 4>{});
    }
    void launch_costas_fast(const cf32 *x, cf32 *y, long long n, const CostasFastParams &p, CostasFastState *state_dev, hipStream_t st)
    { // one chunk = one lane over the whole call; the state is read from and written back to state_dev
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<CostasFastStage>", st);
        ChunkGeom g{};
        g.n = n;
        g.L = 1 << 30;
        g.W = 0;
        g.K = 1;
        auto go = [&](auto stage) {
            using St = decltype(stage);
            hipLaunchKernelGGL((k_chunks<St, false>), dim3(1), dim3(64), 0, st, x, y, g, p, state_dev, state_dev, state_dev, (const int *)nullptr, 0, (CostasFastState *)nullptr, 0, 0, 0.0f,
                               0.0f, (unsigned long long *)nullptr);
        };
        if (p.order == 2)
            go(CostasFastStage<2>{});
        else if (p.order == 4)
            go(CostasFastStage<4>{});
        else
            go(CostasFastStage<8>{});
    }
    void launch_costas_fast_chunks(const cf32 *x, cf32 *y, const ChunkGeom &g, const CostasFastParams &p, const CostasFastState *start0, CostasFastState *spec, CostasFastState *endst,
                                   const int *redo, int nredo, hipStream_t st)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<CostasFastStage>", st);
        auto go = [&](auto stage) {
            using St = decltype(stage);
            hipLaunchKernelGGL((k_chunks<St, false>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (CostasFastState *)nullptr, 0, 0, 0.0f, 0.0f,
                               (unsigned long long *)nullptr);
        };
        if (p.order == 2)
            go(CostasFastStage<2>{});
        else if (p.order == 4)
            go(CostasFastStage<4>{});
        else
            go(CostasFastStage<8>{});
    }
    void launch_pll(const cf32 *x, cf32 *y, const ChunkGeom &g, const PllParams &p, const CostasState *start0, CostasState *spec, CostasState *endst,
                    const int *redo, int nredo, hipStream_t st, const ChunkCkpt &ck)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_chunks<PllStage>", st);
        if (ck.ck)
            hipLaunchKernelGGL((k_chunks<PllStage, true>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo, (CostasState *)ck.ck,
                               ck.per_chunk, ck.len, ck.tol_a, ck.tol_b, ck.work);
        else
            hipLaunchKernelGGL((k_chunks<PllStage, false>), dim3((n + 63) / 64), dim3(64), 0, st, x, y, g, p, start0, spec, endst, redo, nredo,
                               (CostasState *)nullptr, 0, 0, 0.0f, 0.0f, (unsigned long long *)nullptr);
    }

    // =============================================================================================
    // M&M clock recovery. Input = Costas output de-rotated per Costas chunk (+ OQPSK half-symbol delay).
    // =============================================================================================
    __device__ __forceinline__ cf32 rot_apply(cf32 v, int q, int order)
    {
        // multiply by exp(+j*q*unit): unit = pi/2 (order 4), pi (order 2), pi/4 (order 8)
        if (order == 2)
        {
            if (q & 1)
            {
                v.re = -v.re;
                v.im = -v.im;
            }
            return v;
        }
        int quarter = q, eighth = 0;
        if (order == 8)
        {
            quarter = q >> 1;
            eighth = q & 1;
        }
        switch (quarter & 3)
        {
        case 1:
        {
            const float t = v.re;
            v.re = -v.im;
            v.im = t;
            break;
        }
        case 2:
            v.re = -v.re;
            v.im = -v.im;
            break;
        case 3:
        {
            const float t = v.re;
            v.re = v.im;
            v.im = -t;
            break;
        }
        default:
            break;
        }
        if (eighth)
        {
            const float c = 0.70710678118654752f;
            const float r = (v.re - v.im) * c, i2 = (v.re + v.im) * c;
            v.re = r;
            v.im = i2;
        }
        return v;
    }
    __device__ __forceinline__ int costas_chunk_of(const ChunkGeom &g, long long i)
    {
        if (i < (long long)g.L + g.W)
            return 0;
        long long k = (i - g.W) / g.L;
        if (k >= g.K)
            k = g.K - 1;
        return (int)k;
    }
    // ---- per-lane sample window in LDS -------------------------------------------------------------------------
    // A lane walks its own contiguous range of the Costas output. The last MM_RING samples it has fetched live in a
    // private LDS ring (slot = index & (MM_RING-1)). The loop nest is BLOCK-outer / SYMBOL-inner: every outer step moves
    // one 64-byte block (8 samples) from a statically named register queue MM_DEPTH blocks deep into the ring -- applying
    // the per-Costas-chunk rotation and the OQPSK one-sample Q delay (delay_one_imag.cpp:20-27) on the way -- re-issues
    // that queue slot's global loads for the block MM_DEPTH further on, and then runs every M&M iteration whose 8-tap
    // window is now complete. The queue never shifts (a register move would have to wait for its load), so the compiler
    // can count outstanding loads exactly and the ~1.3 us HBM latency of these chunk-strided reads stays off the timing
    // recurrence; the interpolator reads samples and taps from LDS with lane-private addresses.
    constexpr int MM_RING = 32;    // Gardner lanes (their window reaches up to MM_BACK_MAX samples further back)
    constexpr int MM_RING_MM = 16; // Mueller & Mueller lanes: a window [inc-7, inc] with inc in the newest block lies inside the last 16 samples; 11.5 KB of ring
                                   // + 4 KB of interpolator arms per wave lets a CU hold eight blocks -- two waves on every SIMD (it was six: one and a half)
    // ring[slot][lane]: slot-major, so lane l of a 32-lane LDS access group always owns 8-byte bank pair l whatever slot it
    // addresses -- the lanes' windows sit at unrelated ring positions, and a lane-major layout made them collide at random
    // (SQ: bank-conflict cycles were twice the active LDS cycles)
    constexpr int MM_RING_STRIDE = 64; // cf32 units between consecutive slots of one lane
    // Mirror region: whenever slots 0..6 are written they are written a second time at MM_RING..MM_RING+6, so an 8-tap window that starts at any slot of
    // the ring is 8 CONSECUTIVE slots [base, base+7] -- one address per symbol and immediate offsets instead of a wrap per tap (the ring addresses were
    // ~24 of the ~130 VALU instructions of a symbol). A window lies inside the last MM_RING samples, so the copy a reader finds behind slot 31 is the
    // newest write of that slot, the very value the wrapped read returned.
    constexpr int MM_MIRROR = 7;
    constexpr int MM_ARM_STRIDE = 8; // floats between interpolator arms in the LDS copy (a 48-byte stride was measured in round 4: no change)
    constexpr int MM_DEPTH = 4;
    __device__ __forceinline__ void mm_rot_cs(int q, int order, float &c, float &s)
    {
        // exp(+j*q*2pi/order) on the eighth-turn grid
        const int e = (q * (8 / order)) & 7;
        const float h = 0.70710678118654752f;
        c = (e == 0) ? 1.0f : (e == 4) ? -1.0f : (e == 2 || e == 6) ? 0.0f : (e == 1 || e == 7) ? h : -h;
        s = (e == 2) ? 1.0f : (e == 6) ? -1.0f : (e == 0 || e == 4) ? 0.0f : (e == 1 || e == 3) ? h : -h;
    }
    struct MmFeed
    {
        cf32 *ring;     // this lane's row
        long long next; // first sample index not yet in the ring (multiple of 8)
        long long cend; // first sample index of the NEXT Costas chunk (rotation changes there; multiple of 8)
        int ck;         // Costas chunk of block `next`
        int rot_nx;     // rot[ck + 1], loaded when the lane entered chunk ck: the load is long back when the lane gets there (a load issued AT
                        // the crossing stalled the whole wave for a memory round trip, once per lane and Costas chunk)
        float rc, rs;   // its rotation exp(+j*rot*unit) as (cos, sin): exactly 0 / +-1 for quarter and half turns
        float prev_im;  // OQPSK: imaginary part of sample next-1 (after rotation)
    };
    // move one block into the ring. Stage chunk boundaries and 0 are multiples of 8, so a block never straddles a rotation
    // change or the history/data boundary.
    template <int RING = MM_RING>
    __device__ __forceinline__ void mm_feed_put(MmFeed &f, const MmParams &p, const Blk8 &c)
    {
        const long long i = f.next;
        if (p.rot && i >= f.cend && f.ck + 1 < p.cg.K)
        {
            f.ck++;
            f.cend += p.cg.L;
            mm_rot_cs(f.rot_nx, p.order, f.rc, f.rs);
            f.rot_nx = p.rot[f.ck + 1 < p.cg.K ? f.ck + 1 : f.ck];
        }
        const bool dorot = p.rot && i >= 0; // history (negative indices) was rotated by the previous call
        float re[8] = {c.a.x, c.a.z, c.b.x, c.b.z, c.c.x, c.c.z, c.d.x, c.d.z};
        float im[8] = {c.a.y, c.a.w, c.b.y, c.b.w, c.c.y, c.c.w, c.d.y, c.d.w};
        const int slot = (int)(i & (RING - 1));
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            cf32 v{re[j], im[j]};
            if (dorot) // branch-free: multiplications by 0 / +-1 are exact (only the sign of a zero can differ from a swap)
                v = cf32{v.re * f.rc - v.im * f.rs, v.re * f.rs + v.im * f.rc};
            if (p.oqpsk)
            {
                const float t = v.im;
                v.im = f.prev_im;
                f.prev_im = t;
            }
            f.ring[(slot + j) * MM_RING_STRIDE] = v;
            if (j < MM_MIRROR && slot == 0)
                f.ring[(RING + j) * MM_RING_STRIDE] = v;
        }
        f.next = i + 8;
    }
    // start the window so that samples [inc-7-back, inc] can be served
    __device__ __forceinline__ void mm_feed_init(MmFeed &f, const MmParams &p, const cf32 *x, cf32 *ring, long long inc)
    {
        f.ring = ring;
        long long first = inc - 7 - p.back; // Gardner: the zero-crossing window lies up to p.back samples behind the symbol's
        first = (first >= 0 ? first : first - 15) / 16 * 16; // floor to a multiple of 16 (also for negative indices): whole 128-byte bursts from the chunk's first block on
        f.next = first;
        f.ck = 0;
        f.rot_nx = 0;
        f.rc = 1.0f;
        f.rs = 0.0f;
        f.cend = 0;
        if (p.rot)
        {
            f.ck = first >= 0 ? costas_chunk_of(p.cg, first) : 0;
            mm_rot_cs(p.rot[f.ck], p.order, f.rc, f.rs);
            f.rot_nx = p.rot[f.ck + 1 < p.cg.K ? f.ck + 1 : f.ck];
            f.cend = chunk_end(p.cg, f.ck);
        }
        f.prev_im = 0.0f;
        if (p.oqpsk)
        {
            // samples at negative indices are the carried history, already rotated by the previous call
            cf32 v = x[first - 1];
            if (first - 1 >= 0 && p.rot)
                v = rot_apply(v, p.rot[costas_chunk_of(p.cg, first - 1)], p.order);
            f.prev_im = v.im;
        }
    }

    // one iteration of MMClockRecoveryBlock<complex_t>::work's loop body, clock_recovery_mm.cpp:54-120; the window
    // [inc-7, inc] must be in the ring
    // the body of ndsp::MMClockRecoveryFastBlock<complex_t>::work's loop behind its delay-line shift (dsp/clock_recovery/clock_recovery_mm_fast.cpp:108-150), x0 / x1 = the
    // block's buffer[inc] / buffer[inc + 1] = samples inc - 7 / inc - 6 of the stream
    __device__ __forceinline__ cf32 mmfast_core(MmState &s, const MmParams &p, const cf32 x0, const cf32 x1, const float omega_gain, const float mu_gain)
    {
        const float w0 = (float)(1.0 - (double)s.mu);
        const float re = (x0.re * w0) + (x1.re * s.mu), im = (x0.im * w0) + (x1.im * s.mu);
        s.p_0T.re = re;
        s.p_0T.im = im;
        s.c_0T.re = re > 0.0f ? 1.0f : 0.0f;
        s.c_0T.im = im > 0.0f ? 1.0f : 0.0f;
        const float ur = s.p_0T.re - s.p_2T.re, ui = s.p_0T.im - s.p_2T.im;
        const float a_re = (ur * s.c_1T.re) - (ui * (-s.c_1T.im));
        const float vr = s.c_0T.re - s.c_2T.re, vi = s.c_0T.im - s.c_2T.im;
        const float b_re = (vr * s.p_1T.re) - (vi * (-s.p_1T.im));
        float pe = a_re - b_re;
        pe = pe < -1.0f ? -1.0f : (pe > 1.0f ? 1.0f : pe);
        const cf32 out = s.p_0T;
        if (s.upd_cnt++ == 4u)
        {
            s.upd_cnt = 0;
            s.omega = s.omega + omega_gain * pe;
            float d = s.omega - p.omega_mid;
            d = d < -p.omega_limit ? -p.omega_limit : (d > p.omega_limit ? p.omega_limit : d);
            s.omega = p.omega_mid + d;
        }
        s.mu = (s.mu + s.omega) + mu_gain * pe;
        const float fl = floorf(s.mu);
        s.inc += (long long)(int)fl;
        s.mu = s.mu - fl;
        if (s.inc < 0)
            s.inc = 0;
        return out;
    }

    // LIN: ndsp::MMClockRecoveryFastBlock<complex_t>::work's loop body instead (dsp/clock_recovery/clock_recovery_mm_fast.cpp:96-150): the symbol is the linear
    // interpolation buffer[inc] * (1.0 - mu) + buffer[inc + 1] * mu -- samples inc - 7 and inc - 6 of the stream, the block's buffer holding ntaps - 1 = 7 samples of
    // history in front; (1.0 - mu) taken in double and rounded to the float complex_t::operator*(const float &) takes -- and the rate term moves on every fifth symbol
    template <bool FAST = false, bool TAP = false, int RING = MM_RING, bool LIN = false>
    __device__ __forceinline__ cf32 mm_iter(MmState &s, const MmParams &p, const cf32 *ring, const float *bank, const float omega_gain, const float mu_gain,
                                            long long *arm_pos = nullptr)
    {
        s.p_2T = s.p_1T;
        s.p_1T = s.p_0T;
        s.c_2T = s.c_1T;
        s.c_1T = s.c_0T;
        if constexpr (LIN)
        {
            const int b0 = (int)((s.inc - 7) & (RING - 1)), b1 = (int)((s.inc - 6) & (RING - 1));
            return mmfast_core(s, p, ring[b0 * MM_RING_STRIDE], ring[b1 * MM_RING_STRIDE], omega_gain, mu_gain);
        }
        int imu = (int)rintf(s.mu * 128.0f);
        if (imu < 0)
            imu = 0;
        if (imu >= 128)
            imu = 127;
        if constexpr (TAP)
            *arm_pos = s.inc * 128 + imu;
        const float4 t0 = *reinterpret_cast<const float4 *>(bank + imu * MM_ARM_STRIDE);
        const float4 t1 = *reinterpret_cast<const float4 *>(bank + imu * MM_ARM_STRIDE + 4);
        const float t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        const int base = (int)((s.inc - 7) & (RING - 1));
        // (re, im) of a sample as one packed pair: v_pk_mul_f32 / v_pk_add_f32, each half rounded like the scalar operation
        v2f acc{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const cf32 v = ring[(base + k) * MM_RING_STRIDE];
            if constexpr (FAST)
                acc = __builtin_elementwise_fma(v2f{v.re, v.im}, v2f{t[k], t[k]}, acc); // chunk-parallel mode's arithmetic (see sd_sincosf_fast)
            else
            {
                const v2f prod = v2f{v.re, v.im} * v2f{t[k], t[k]};
                acc = acc + prod;
            }
        }
        const float re = acc.x, im = acc.y;
        s.p_0T.re = re;
        s.p_0T.im = im;
        s.c_0T.re = re > 0.0f ? 1.0f : 0.0f;
        s.c_0T.im = im > 0.0f ? 1.0f : 0.0f;
        const float ur = s.p_0T.re - s.p_2T.re, ui = s.p_0T.im - s.p_2T.im;
        const float a_re = (ur * s.c_1T.re) - (ui * (-s.c_1T.im));
        const float vr = s.c_0T.re - s.c_2T.re, vi = s.c_0T.im - s.c_2T.im;
        const float b_re = (vr * s.p_1T.re) - (vi * (-s.p_1T.im));
        float pe = a_re - b_re;
        pe = pe < -1.0f ? -1.0f : (pe > 1.0f ? 1.0f : pe); // branched_clip(phase_error, 1.0)
        const cf32 out = s.p_0T;
        s.omega = s.omega + omega_gain * pe;
        float d = s.omega - p.omega_mid;
        d = d < -p.omega_limit ? -p.omega_limit : (d > p.omega_limit ? p.omega_limit : d);
        s.omega = p.omega_mid + d;
        s.mu = (s.mu + s.omega) + mu_gain * pe;
        const float fl = floorf(s.mu);
        s.inc += (long long)(int)fl;
        s.mu = s.mu - fl;
        if (s.inc < 0)
            s.inc = 0;
        return out;
    }

    // one iteration of GardnerClockRecoveryBlock<complex_t>::work's loop body (legacy: common/dsp/clock_recovery/clock_recovery_gardner.cpp:49-106;
    // ndsp: dsp/clock_recovery/clock_recovery_gardner.cpp:88-138 -- the same statements, its two clips on floats): two interpolations per symbol -- the
    // symbol itself over [inc-7, inc] and the zero crossing half a symbol back over [inc-offzc-7, inc-offzc] -- the float / double promotions where C++
    // puts them. The window [inc-7-p.back, inc] must be in the ring; the last symbol rides in s.p_0T.
    template <bool FAST = false>
    __device__ __forceinline__ cf32 gardner_iter(MmState &s, const MmParams &p, const cf32 *ring, const float *bank, const float omega_gain, const float mu_gain)
    {
        const float muz = (float)((double)s.mu - ((double)s.omega / 2.0));
        int offzc = (int)floor((double)s.omega / 2.0);
        float mupos = (float)fmod((double)(muz + (float)offzc), 1.0);
        if (mupos < 0)
        {
            mupos = 1 + mupos;
            offzc += 1;
        }
        int imuz = (int)rint((double)(mupos * 128.0f));
        imuz = imuz < 0 ? 0 : (imuz >= 128 ? 127 : imuz);
        int imu = (int)rint((double)(s.mu * 128.0f));
        imu = imu < 0 ? 0 : (imu >= 128 ? 127 : imu);
        if (offzc > p.back) // cannot happen inside the omega limits the engine admits; keeps a wild state inside the ring
            offzc = p.back;
        const float4 z0 = *reinterpret_cast<const float4 *>(bank + imuz * MM_ARM_STRIDE), z1 = *reinterpret_cast<const float4 *>(bank + imuz * MM_ARM_STRIDE + 4);
        const float4 t0 = *reinterpret_cast<const float4 *>(bank + imu * MM_ARM_STRIDE), t1 = *reinterpret_cast<const float4 *>(bank + imu * MM_ARM_STRIDE + 4);
        const float tz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
        const float t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        const int base = (int)((s.inc - 7) & (MM_RING - 1)), basez = (int)((s.inc - offzc - 7) & (MM_RING - 1));
        v2f az{0.0f, 0.0f}, as{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const cf32 v = ring[(basez + k) * MM_RING_STRIDE];
            if constexpr (FAST)
                az = __builtin_elementwise_fma(v2f{v.re, v.im}, v2f{tz[k], tz[k]}, az);
            else
            {
                const v2f prod = v2f{v.re, v.im} * v2f{tz[k], tz[k]};
                az = az + prod;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const cf32 v = ring[(base + k) * MM_RING_STRIDE];
            if constexpr (FAST)
                as = __builtin_elementwise_fma(v2f{v.re, v.im}, v2f{t[k], t[k]}, as);
            else
            {
                const v2f prod = v2f{v.re, v.im} * v2f{t[k], t[k]};
                as = as + prod;
            }
        }
        const float zr = az.x, zi = az.y, sr = as.x, si = as.y;
        float pe = zr * (s.p_0T.re - sr) + zi * (s.p_0T.im - si);
        if (p.clip_float)
            pe = pe < -1.0f ? -1.0f : (pe > 1.0f ? 1.0f : pe);
        else
            pe = (float)(0.5 * (fabs((double)pe + 1.0) - fabs((double)pe - 1.0)));
        s.p_0T = cf32{sr, si};
        s.omega = s.omega + omega_gain * pe;
        const float d = s.omega - p.omega_mid;
        if (p.clip_float)
            s.omega = p.omega_mid + (d < -p.omega_limit ? -p.omega_limit : (d > p.omega_limit ? p.omega_limit : d));
        else
            s.omega = (float)((double)p.omega_mid + 0.5 * (double)(fabsf(d + p.omega_limit) - fabsf(d - p.omega_limit)));
        s.mu = s.mu + s.omega + mu_gain * pe;
        const float fl = floorf(s.mu);
        s.inc += (long long)(int)fl;
        s.mu = s.mu - fl;
        if (s.inc < 0)
            s.inc = 0;
        return s.p_0T;
    }
    template <bool GARD, bool FAST, bool TAP = false, int RING = MM_RING, bool LIN = false>
    __device__ __forceinline__ cf32 clock_iter(MmState &s, const MmParams &p, const cf32 *ring, const float *bank, const float omega_gain, const float mu_gain)
    {
        if constexpr (LIN)
            return mm_iter<false, false, RING, true>(s, p, ring, bank, omega_gain, mu_gain);
        else if constexpr (GARD)
            return gardner_iter<FAST>(s, p, ring, bank, omega_gain, mu_gain);
        else if constexpr (TAP)
        { // tests only: the symbol's eight bytes carry its position on the arm grid (MmParams::tap)
            long long pos = 0;
            (void)mm_iter<FAST, true, RING>(s, p, ring, bank, omega_gain, mu_gain, &pos);
            return cf32{__uint_as_float((unsigned)(pos & 0xffffffffll)), __uint_as_float((unsigned)((unsigned long long)pos >> 32))};
        }
        else
            return mm_iter<FAST, false, RING>(s, p, ring, bank, omega_gain, mu_gain);
    }

    // CKPT: whenever the block ending at a multiple of MM_CK_SAMPLES samples into its chunk has been fed, a lane leaves a
    // checkpoint {mu, omega, inc, symbols so far}. A re-run lane (exact start state) compares itself with the checkpoint at the
    // same position and stops as soon as it has produced the same number of symbols and is inside the boundary tolerance in
    // time: from there on the speculative output, count and end state of the chunk stand under the very rule that accepts a
    // chunk boundary. A lane that does not merge overwrites the checkpoints, so they always describe the trajectory whose
    // symbols are in the scratch rows. (The test sits in the block loop, not in the symbol loop: a checkpoint per symbol count
    // cost the main launch 12 % -- measured, MetOp: 14.8 against 13.2 ms.)
    // SPLIT (experimental, SDHIP_MM_SPLIT=1, same results bit for bit -- checked on the host twin; not yet measured): the symbol loop
    // as three plain loops, one per phase, each bounded by a single sample-index test, instead of one loop that re-evaluates the
    // warm-up / chunk / look-ahead bookkeeping (~45 of its ~200 instructions) on every symbol. The lane is issue-bound.
        // quantiser, module_psk_demod.cpp:199-213 + clamp module_demod_base.h:106-113
    __device__ __forceinline__ signed char sd_clamp8(float x)
    {
        if (x < -128.0f)
            return -127;
        if (x > 127.0f)
            return 127;
        return (signed char)(int)x;
    }
    // the same value without branches (the symbol loop of k_mm<.., Q8>): inside [-128, 127] the conversion truncates as above, beyond 127 the clamp yields 127,
    // below -128 the select puts -127 where the clamp left -128
    __device__ __forceinline__ unsigned sd_clamp8_u(float x)
    {
        const int r = (int)fminf(fmaxf(x, -128.0f), 127.0f);
        return (unsigned)(x < -128.0f ? -127 : r) & 0xffu;
    }
template <bool CKPT, bool SPLIT, bool Q8 = false, bool FAST = false, bool GARD = false, bool TAP = false, bool LIN = false>
    __global__ __launch_bounds__(64) void k_mm(const cf32 *x, cf32 *sym, int *counts, ChunkGeom g, MmParams p, const MmState *start0, MmState *spec,
                                               MmState *endst, MmCert *spec_c, MmCert *end_c, const int *redo, int nredo, MmCkpt *ck, int ck_per_chunk,
                                               float ck_tol, int coop_nb)
    {
        constexpr int RING = GARD ? MM_RING : MM_RING_MM;
        __shared__ cf32 rings[(RING + MM_MIRROR) * MM_RING_STRIDE];
        __shared__ __attribute__((aligned(16))) float bank[128 * MM_ARM_STRIDE];
        // the transposition buffer of the cooperative loads (load side only: the symbols leave per lane) is DYNAMIC shared memory, there only when a launch uses it
        // (SDHIP_COOP_MM=1; off by default): the block's static LDS is ring + bank = 15.5 KB, ten blocks to a CU (it was 24.5 KB, six)
#ifdef SDHIP_HOST_TWIN
        static __attribute__((aligned(16))) char coop_lds[COOP_LDS_BYTES];
#else
        extern __shared__ __attribute__((aligned(16))) char coop_lds[];
#endif
        for (int i = (int)threadIdx.x; i < 128 * 8; i += 64)
            bank[(i >> 3) * MM_ARM_STRIDE + (i & 7)] = p.bank[i];
        __syncthreads();
        const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        // cooperative loads (see Coop, k_afc): blocks 0 .. coop_nb-1 of a main launch take the chunks 1 + 64 blk + lane, chunk 0 and the tail follow eight to a block
        const bool coop = !redo && coop_nb > 0 && (int)blockIdx.x < coop_nb;
        int k;
        MmState s;
        bool warm = false;
        if (redo)
        {
            if (idx >= nredo)
                return;
            k = redo[idx];
            s = spec[k]; // snapshot of endst[k-1] taken before the launch (see k_chunks)
        }
        else
        {
            if (coop_nb > 0)
            {
                if (coop)
                    k = 1 + idx;
                else
                {
                    const int j = ((int)blockIdx.x - coop_nb) * 8 + (int)threadIdx.x;
                    if ((int)threadIdx.x >= 8 || j >= g.K - 64 * coop_nb)
                        return;
                    k = j == 0 ? 0 : 64 * coop_nb + j;
                }
            }
            else
            {
                k = idx;
                if (k >= g.K)
                    return;
            }
            if (k == 0)
                s = *start0;
            else
            {
                s.mu = p.init_mu;
                s.omega = p.omega_mid;
                s.p_2T = s.p_1T = s.p_0T = cf32{0.0f, 0.0f};
                s.c_2T = s.c_1T = s.c_0T = cf32{0.0f, 0.0f};
                s.inc = chunk_begin(g, k) - g.W;
                s.upd_cnt = 0;
                s.pad = 0;
                warm = true;
            }
        }
        MmFeed f;
        mm_feed_init(f, p, x, rings + (int)threadIdx.x, s.inc);
        // Three phases: 0 = warm-up (nothing stored) until the chunk start, 1 = the chunk itself, 2 = up to two look-ahead
        // symbols past the chunk end computed from the running state AFTER the end state has been saved: if the next
        // chunk's own trajectory starts one symbol later than this one ends (timing within tolerance, boundary sample
        // index on the other side of the mu wrap), the host hands these to the stream instead of re-running anything.
        const long long b = chunk_begin(g, k), e = chunk_end(g, k);
        // Q8: the symbols leave as the module's int8 soft symbols (x100, x50 for BPSK, clamped: module_psk_demod.cpp:199-213) -- two
        // bytes per symbol in the scratch row instead of eight; the float symbols are only stored when a caller asks for them
        cf32 *o = sym + (size_t)k * p.cap;
        short *o8 = reinterpret_cast<short *>(sym) + (size_t)k * p.cap;
        // (eight symbols leave as ONE aligned 16-byte store: a lane's symbols get consecutive indices from 0 -- the chunk's, then its look-ahead's --, the rows start
        // on 16-byte boundaries (the engine rounds the row length to eight symbols); the newest eight sit in a 128-bit shift register, the newest in the top
        // halfword, and what is left of a group when the lane is done goes out halfword by halfword)
        unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        int qn = 0, q_last = -1;
        auto q8_flush_part = [&]() {
            const unsigned w[4] = {q0, q1, q2, q3};
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j >= 8 - qn)
                    o8[q_last - 7 + j] = (short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
            qn = 0;
        };
        auto put = [&](int at, const cf32 v) {
            if constexpr (Q8)
            {
                const float sc = p.q8_bpsk ? 50.0f : 100.0f;
                const unsigned cur = sd_clamp8_u(v.re * sc) | (sd_clamp8_u(v.im * sc) << 8);
                q0 = (q0 >> 16) | (q1 << 16);
                q1 = (q1 >> 16) | (q2 << 16);
                q2 = (q2 >> 16) | (q3 << 16);
                q3 = (q3 >> 16) | (cur << 16);
                qn++;
                q_last = at;
                if (qn == 8)
                { // (a lane's first index is 0, so `at - 7` is a multiple of eight and the store aligned; it would be correct, only slower, from any other start)
                    *reinterpret_cast<uint4 *>(o8 + at - 7) = uint4{q0, q1, q2, q3};
                    qn = 0;
                }
            }
            else
                o[at] = v;
        };
        int phase = warm ? 0 : 1, cnt = 0, nx = 0, wsym = 0;
        bool done = false, merged = false;
        // one 8-sample block: into the ring, then every symbol whose window is complete (the body of both loops below)
        const auto block_body = [&](const Blk8 &cur) __attribute__((always_inline)) {
                mm_feed_put<RING>(f, p, cur);
                if constexpr (CKPT)
                { // one test per 8-sample block, outside the symbol loop: a checkpoint every MM_CK_SAMPLES samples of the chunk
                    const long long rel = f.next - b;
                    if (phase == 1 && rel > 0 && (rel & (MM_CK_SAMPLES - 1)) == 0 && f.next < e)
                    {
                        const int j = (int)(rel / MM_CK_SAMPLES) - 1;
                        if (j < ck_per_chunk)
                        {
                            MmCkpt *c = ck + (size_t)k * ck_per_chunk + j;
                            if (redo)
                            {
                                const MmCkpt o = *c;
                                const double dt = (double)(s.inc - o.inc) + ((double)s.mu - (double)o.mu);
                                if (cnt == o.cnt && fabs(dt) < (double)ck_tol && fabsf(s.omega - o.omega) < p.tol_omega)
                                    merged = done = true;
                            }
                            if (!merged)
                                *c = MmCkpt{s.mu, s.omega, s.inc, cnt, 0};
                        }
                    }
                }
                if constexpr (SPLIT && !CKPT)
                {
                    while (!done && s.inc < f.next)
                    {
                        if (phase == 0)
                        {
                            if (s.inc >= b)
                            {
                                spec[k] = s;
                                spec_c[k] = MmCert{s.mu, s.omega, s.inc};
                                phase = 1;
                                continue;
                            }
                            const long long lim = f.next < b ? f.next : b;
                            do
                            { // warm-up symbols: nothing stored (gear shift: see below)
                                const bool fast = wsym < p.fast_syms;
                                wsym++;
                                (void)mm_iter<false, false, RING>(s, p, f.ring, bank, fast ? 0.0f : p.omega_gain, fast ? p.mu_gain * p.fast_mult : p.mu_gain);
                            } while (s.inc < lim);
                        }
                        else if (phase == 1)
                        {
                            if (s.inc >= e)
                            {
                                counts[2 * k] = cnt;
                                endst[k] = s;
                                end_c[k] = MmCert{s.mu, s.omega, s.inc};
                                phase = 2;
                                if (k + 1 >= g.K)
                                    done = true;
                                continue;
                            }
                            const long long lim = f.next < e ? f.next : e;
                            do
                            {
                                const cf32 v = mm_iter<false, false, RING>(s, p, f.ring, bank, p.omega_gain, p.mu_gain);
                                if (cnt < p.cap)
                                    put(cnt, v);
                                cnt++;
                            } while (s.inc < lim);
                        }
                        else
                        {
                            if (nx >= 2 || s.inc >= g.n)
                            {
                                done = true;
                                break;
                            }
                            const cf32 v = mm_iter<false, false, RING>(s, p, f.ring, bank, p.omega_gain, p.mu_gain);
                            if (cnt + nx < p.cap)
                                put(cnt + nx, v);
                            nx++;
                        }
                    }
                    return;
                }
                if constexpr (!SPLIT)
                {
                    // Wave-uniform fast paths. The lanes of a wave sit at the same place relative to their chunks (chunk starts, lengths and warm-ups are the
                    // same multiples of 8), so for all but a handful of the blocks of a chunk EVERY lane is in its warm-up for the whole block, or every lane is
                    // inside its chunk for the whole block: then the symbol loop needs none of the per-symbol phase bookkeeping below (a third of its
                    // instructions; the kernel is issue-bound at two waves per SIMD). Same iterations, same order, same results.
                    const unsigned long long act = __ballot(1);
                    if (__ballot(!done && phase == 1 && f.next <= e && cnt + 8 < p.cap) == act) // (!done: a re-run lane that merged at a checkpoint stays out)
                    {
                        while (s.inc < f.next)
                        {
                            const cf32 v = clock_iter<GARD, FAST, TAP, RING, LIN>(s, p, f.ring, bank, p.omega_gain, p.mu_gain);
                            put(cnt, v);
                            cnt++;
                        }
                        return;
                    }
                    if (__ballot(!done && phase == 0 && f.next <= b) == act)
                    {
                        while (s.inc < f.next)
                        {
                            const bool fast = wsym < p.fast_syms;
                            wsym++;
                            (void)clock_iter<GARD, FAST, TAP, RING, LIN>(s, p, f.ring, bank, fast ? 0.0f : p.omega_gain, fast ? p.mu_gain * p.fast_mult : p.mu_gain);
                        }
                        return;
                    }
                }
                while (!done && s.inc < f.next)
                {
                    if (phase == 0 && s.inc >= b)
                    {
                        spec[k] = s;
                        spec_c[k] = MmCert{s.mu, s.omega, s.inc};
                        phase = 1;
                    }
                    if (phase == 1 && s.inc >= e)
                    {
                        counts[2 * k] = cnt;
                        endst[k] = s;
                        end_c[k] = MmCert{s.mu, s.omega, s.inc};
                        phase = 2;
                        if (k + 1 >= g.K)
                            done = true;
                    }
                    if (phase == 2 && (nx >= 2 || s.inc >= g.n))
                        done = true;
                    if (!done)
                    {
                        // warm-up gear shift: the first fast_syms symbols of a warm-up run with the timing gain raised and the
                        // rate term frozen (pull-in in ~1/fast_mult of the time), the rest with the loop's own gains so that the
                        // trajectory settles onto the sequential one; only speculation -- the boundary certificate decides
                        const bool fast = phase == 0 && wsym < p.fast_syms;
                        wsym++;
                        const cf32 v = clock_iter<GARD, FAST, TAP, RING, LIN>(s, p, f.ring, bank, fast ? 0.0f : p.omega_gain, fast ? p.mu_gain * p.fast_mult : p.mu_gain);
                        if (phase != 0)
                        {
                            if (cnt + nx < p.cap)
                                put(cnt + nx, v);
                            if (phase == 1)
                                cnt++;
                            else
                                nx++;
                        }
                    }
                }
        };
        if (coop)
        { // Cooperative phase: the lanes of this wave are ordinary chunks one chunk length apart and walk in lockstep (same f.next relative to the chunk start, a
          // multiple of 16), nobody is done before its chunk end; bursts of 16 samples while they lie in front of the chunk end, the rest on the per-lane queue below
            const Coop co = coop_make(coop_lds, (long long)g.L * (long long)sizeof(cf32));
            const long long span = e - f.next;
            const int nb = sd_uniform((int)(span > 0 ? span / 16 : 0));
            long long fw = sd_uniform(f.next);
            if (nb > 0)
            {
                Burst b0 = coop_load(x, fw, co), b1 = b0;
                if (nb > 1)
                    b1 = coop_load(x, fw + 16, co);
                for (int it = 0; it < nb; it++)
                {
                    Blk8 c0, c1;
                    coop_unpack(b0, co, c0, c1);
                    b0 = b1;
                    if (it + 2 < nb)
                        b1 = coop_load(x, fw + 32, co);
                    fw += 16;
                    block_body(c0);
                    block_body(c1);
                }
            }
        }
        Blk8 q[MM_DEPTH];
#pragma unroll
        for (int d = 0; d < MM_DEPTH; d++)
            q[d] = blk_load(x, f.next + 8 * d);
        while (!done)
        {
#pragma unroll
            for (int d = 0; d < MM_DEPTH; d++)
            {
                const Blk8 cur = q[d];
                q[d] = blk_load(x, f.next + 8 * MM_DEPTH); // at most 8*MM_DEPTH + 8 samples past the lane's last window
                block_body(cur);
            }
        }
        if constexpr (Q8)
            if (qn > 0)
                q8_flush_part();
        if (!merged) // a merged re-run leaves the chunk's count, look-ahead and end state as the speculative run wrote them
            counts[2 * k + 1] = nx;
    }
    void launch_mm(const cf32 *x, cf32 *sym_scratch, int *counts, const ChunkGeom &g, const MmParams &p, const MmState *start0, MmState *spec, MmState *endst,
                   MmCert *spec_c, MmCert *end_c, const int *redo, int nredo, hipStream_t st, MmCkpt *ck, int ck_per_chunk, float ck_tol)
    {
        const int n = redo ? nredo : g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_mm", st);
        std::optional<ProfScope> _pr; // the re-run launches (a few lanes, each alone on its SIMD) are part of k_mm's time; listed on their own as well
        if (redo)
            _pr.emplace("k_mm (re-run launches, included in k_mm)", st);
        const char *split_env = getenv("SDHIP_MM_SPLIT");
        const bool split = split_env && split_env[0] == '1';
        // cooperative loads of the wave's 64 streams (see Coop): chunk length and warm-up whole 128-byte bursts, at least one full wave of ordinary chunks
        // (measured, visit D of round 5: on these lanes -- latency-bound, one wave per SIMD -- the transposes cost more than the coalescing returns: MetOp 14.6 ms
        // with, 13.3 without at 98 304 lanes; SDHIP_COOP_MM=1 turns it on, tests run both)
        const bool coop_env = !(getenv("SDHIP_COOP") && atoi(getenv("SDHIP_COOP")) == 0) && getenv("SDHIP_COOP_MM") && atoi(getenv("SDHIP_COOP_MM")) != 0;
        int coop_nb = 0;
        if (coop_env && !redo && g.L % 16 == 0 && g.W % 16 == 0 && g.K >= 66)
            coop_nb = (g.K - 2) / 64;
        if (!redo && coop_nb == 0 && g.K >= 66 && getenv("SDHIP_COOP_REQUIRE"))
            throw HipError("k_mm: cooperative access asked for (SDHIP_COOP_REQUIRE) but the geometry does not allow it");
        const int nblk = coop_nb > 0 ? coop_nb + (g.K - 64 * coop_nb + 7) / 8 : (n + 63) / 64;
        if (getenv("SDHIP_DEBUG") && !redo)
            fprintf(stderr, "[sdhip] k_mm: K %d L %d W %d -> %d cooperative blocks of %d\n", g.K, g.L, g.W, coop_nb, nblk);
        auto go = [&](auto kern, MmCkpt *ckp, int per, float tol) {
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), coop_nb > 0 ? COOP_LDS_BYTES : 0, st, x, sym_scratch, counts, g, p, start0, spec, endst, spec_c, end_c, redo, nredo, ckp, per,
                               tol, coop_nb);
        };
        if (p.tap)
        { // tests only (sdhip_demod_set_tap): the default instance with the arm positions in place of the symbols
            if (!ck || !p.fast || p.q8 || p.loop == 1)
                throw HipError("the arm tap exists for the chunk-parallel mode's default kernel only");
            go(k_mm<true, false, false, true, false, true>, ck, ck_per_chunk, ck_tol);
            return;
        }
        if (p.loop == 2)
        { // ndsp::MMClockRecoveryFastBlock on ONE sequential lane (float symbols): the cadence of its rate updates follows the symbol count (demod_engine.hip, ndsp_create)
            if (g.K != 1 || p.q8 || p.fast || redo)
                throw HipError("fast_clock_recovery_mm_cc runs as one sequential lane");
            go(k_mm<false, false, false, false, false, false, true>, nullptr, 0, 0.0f);
            return;
        }
        if (p.loop == 1)
        { // the Gardner loop on the same lanes (float symbols only)
            if (p.back < 1 || p.back > MM_BACK_MAX || p.q8)
                throw HipError("Gardner lanes: omega out of the window the lanes carry");
            if (ck && p.fast)
                go(k_mm<true, false, false, true, true>, ck, ck_per_chunk, ck_tol);
            else if (ck)
                go(k_mm<true, false, false, false, true>, ck, ck_per_chunk, ck_tol);
            else if (p.fast)
                go(k_mm<false, false, false, true, true>, nullptr, 0, 0.0f);
            else
                go(k_mm<false, false, false, false, true>, nullptr, 0, 0.0f);
            return;
        }
        if (ck && p.q8 && p.fast)
            go(k_mm<true, false, true, true>, ck, ck_per_chunk, ck_tol);
        else if (ck && p.q8)
            go(k_mm<true, false, true>, ck, ck_per_chunk, ck_tol);
        else if (ck && p.fast)
            go(k_mm<true, false, false, true>, ck, ck_per_chunk, ck_tol);
        else if (ck)
            go(k_mm<true, false>, ck, ck_per_chunk, ck_tol);
        else if (p.q8)
            go(k_mm<false, false, true>, nullptr, 0, 0.0f);
        else if (split)
            go(k_mm<false, true>, nullptr, 0, 0.0f);
        else
            go(k_mm<false, false>, nullptr, 0, 0.0f);
    }


    // ---- fast_clock_recovery_mm_cc lane per (chunk, cadence) ------------------------------------------------------------------------------------------------
    // The block's rate term moves on every fifth SYMBOL (omega_upd_cnt), and how many symbols lie in front of a chunk nobody knows before they have been
    // counted -- so every chunk is run FIVE times, once per value of the counter at its warm-up's start. Lanes of the right cadence merge with the sequential
    // trajectory bit for bit (measured on the reference block itself: 12 - 25 k symbols at the default gains, tools note in DESIGN 7b), and the engine picks for
    // every chunk the variant whose state at the chunk start IS its predecessor's state at its end (DemodEngine::mmfast_stage). Variant-major lane order: a wave
    // holds 64 consecutive chunks of one cadence. A lane reads samples inc - 7 and inc - 6 straight from memory (7 samples of history in front of x).
    // rows: [K][6][cap] symbols; spec / endst / counts: [K][6]: slots 0 - 4 the cadences, slot 5 the re-run (exact start state redo_start[i]) of a chunk none of
    // whose variants stood.
    // (chunk 0 is W + L samples long and runs once: its row, cap0 symbols, lies behind the K x 5 rows of cap symbols)
    __global__ __launch_bounds__(64) void k_mmfast(const cf32 *__restrict__ x, cf32 *rows, int *counts, ChunkGeom g, int cap, int cap0, MmParams p, const MmState *start0, MmState *spec,
                                                   MmState *endst, const int *redo, const MmState *redo_start, int nredo)
    {
        const int idx = (int)(blockIdx.x * 64 + threadIdx.x);
        int k, v;
        MmState s;
        if (redo)
        {
            if (idx >= nredo)
                return;
            k = redo[idx];
            v = 5;
            s = redo_start[idx];
        }
        else
        {
            if (idx >= 5 * g.K)
                return;
            v = idx / g.K;
            k = idx - v * g.K;
            if (k == 0)
            {
                if (v != 0)
                    return;
                s = *start0;
            }
            else
            {
                s.mu = p.init_mu;
                s.omega = p.omega_mid;
                s.p_2T = s.p_1T = s.p_0T = cf32{0.0f, 0.0f};
                s.c_2T = s.c_1T = s.c_0T = cf32{0.0f, 0.0f};
                s.inc = chunk_begin(g, k) - g.W;
                s.upd_cnt = (unsigned)v;
                s.pad = 0;
            }
        }
        auto iter = [&]() {
            s.p_2T = s.p_1T;
            s.p_1T = s.p_0T;
            s.c_2T = s.c_1T;
            s.c_1T = s.c_0T;
            const cf32 x0 = x[s.inc - 7], x1 = x[s.inc - 6];
            return mmfast_core(s, p, x0, x1, p.omega_gain, p.mu_gain);
        };
        const long long b = chunk_begin(g, k), e = chunk_end(g, k);
        const size_t slot = (size_t)k * 6 + (size_t)v;
        if (!redo && k > 0)
        {
            while (s.inc < b)
                (void)iter();
            spec[slot] = s;
        }
        cf32 *row = k == 0 ? rows + (size_t)g.K * 6 * (size_t)cap : rows + slot * (size_t)cap;
        const int room = k == 0 ? cap0 : cap;
        int cnt = 0;
        while (s.inc < e && cnt < room)
            row[cnt++] = iter();
        endst[slot] = s;
        counts[slot] = s.inc < e ? -1 : cnt; // -1: the row overflowed (cannot happen inside the omega limits the engine admits)
    }
    void launch_mmfast(const cf32 *x, cf32 *rows, int *counts, const ChunkGeom &g, int cap, int cap0, const MmParams &p, const MmState *start0, MmState *spec, MmState *endst,
                       const int *redo, const MmState *redo_start, int nredo, hipStream_t st)
    {
        const int n = redo ? nredo : 5 * g.K;
        if (n <= 0)
            return;
        ProfScope _ps("k_mmfast", st);
        hipLaunchKernelGGL(k_mmfast, dim3((n + 63) / 64), dim3(64), 0, st, x, rows, counts, g, cap, cap0, p, start0, spec, endst, redo, redo_start, nredo);
    }
    // the chosen variants' rows, end to end: a block per chunk
    __global__ __launch_bounds__(256) void k_mmfast_gather(const cf32 *__restrict__ rows, const int *__restrict__ sel, const long long *__restrict__ offs, const int *__restrict__ counts, int K,
                                                           int cap, cf32 *out)
    {
        const int k = (int)blockIdx.x;
        if (k >= K)
            return;
        const size_t slot = (size_t)k * 6 + (size_t)sel[k];
        const int cnt = counts[slot];
        const cf32 *row = k == 0 ? rows + (size_t)K * 6 * (size_t)cap : rows + slot * (size_t)cap;
        cf32 *o = out + offs[k];
        for (int i = (int)threadIdx.x; i < cnt; i += 256)
            o[i] = row[i];
    }
    void launch_mmfast_gather(const cf32 *rows, const int *sel, const long long *offs, const int *counts, int K, int cap, cf32 *out, hipStream_t st)
    {
        if (K <= 0)
            return;
        ProfScope _ps("k_mmfast_gather", st);
        hipLaunchKernelGGL(k_mmfast_gather, dim3(K), dim3(256), 0, st, rows, sel, offs, counts, K, cap, out);
    }

    __global__ __launch_bounds__(256) void k_quantize(const cf32 *sym, const int *seg, const long long *offsets, int K, int cap, int bpsk, int8_t *soft,
                                                      long long soft_cap, float *syms, long long syms_cap)
    {
        const int k = (int)blockIdx.x;
        if (k >= K)
            return;
        const int cnt = seg[2 * k + 1];
        const long long off = offsets[k];
        const cf32 *s = sym + (size_t)k * cap + seg[2 * k];
        const uintptr_t oaddr = reinterpret_cast<uintptr_t>(soft) + (uintptr_t)(bpsk ? off : 2 * off); // where this row's bytes go
        if (!syms && off + cnt <= (bpsk ? soft_cap : soft_cap / 2) && (bpsk || (oaddr & 1) == 0))
        {
            // the usual case (nobody asked for the float symbols, the row fits): four symbols per thread and ONE aligned store of
            // their 4 (BPSK) / 8 (QPSK) bytes instead of a byte store per soft symbol; the row's first few symbols, up to the
            // next 4- / 8-byte boundary of the output, and its last few go one by one
            const int to_boundary = bpsk ? (int)((4 - (oaddr & 3)) & 3) : (int)(((8 - (oaddr & 7)) & 7) / 2);
            const int head = to_boundary < cnt ? to_boundary : cnt;
            const int groups = (cnt - head) / 4;
            const float sc = bpsk ? 50.0f : 100.0f;
            auto one = [&](int j) {
                const cf32 v = s[j];
                const long long o = off + j;
                if (bpsk)
                    soft[o] = sd_clamp8(v.re * sc);
                else
                {
                    soft[2 * o] = sd_clamp8(v.re * sc);
                    soft[2 * o + 1] = sd_clamp8(v.im * sc);
                }
            };
            if ((int)threadIdx.x < head)
                one((int)threadIdx.x);
            for (int g = (int)threadIdx.x; g < groups; g += (int)blockDim.x)
            {
                const int j = head + 4 * g;
                unsigned lo = 0, hi = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const cf32 v = s[j + q];
                    const unsigned a = (unsigned)(unsigned char)sd_clamp8(v.re * sc);
                    if (bpsk)
                        lo |= a << (8 * q);
                    else
                    {
                        const unsigned b = (unsigned)(unsigned char)sd_clamp8(v.im * sc);
                        const unsigned pr = a | (b << 8);
                        if (q < 2)
                            lo |= pr << (16 * q);
                        else
                            hi |= pr << (16 * (q - 2));
                    }
                }
                const long long o = off + j;
                if (bpsk)
                    *reinterpret_cast<unsigned *>(soft + o) = lo;
                else
                    *reinterpret_cast<unsigned long long *>(soft + 2 * o) = (unsigned long long)lo | ((unsigned long long)hi << 32);
            }
            const int tail0 = head + 4 * groups;
            if ((int)threadIdx.x < cnt - tail0)
                one(tail0 + (int)threadIdx.x);
            return;
        }
        for (int j = (int)threadIdx.x; j < cnt; j += (int)blockDim.x)
        {
            const cf32 v = s[j];
            const long long o = off + j;
            if (syms && o < syms_cap)
            {
                syms[2 * o] = v.re;
                syms[2 * o + 1] = v.im;
            }
            if (bpsk)
            {
                if (o < soft_cap)
                    soft[o] = sd_clamp8(v.re * 50.0f);
            }
            else if (2 * o + 1 < soft_cap)
            {
                soft[2 * o] = sd_clamp8(v.re * 100.0f);
                soft[2 * o + 1] = sd_clamp8(v.im * 100.0f);
            }
        }
    }
    // compaction of the int8 scratch rows k_mm<.., Q8> leaves (two bytes per symbol): BPSK keeps the first byte of each pair.
    // Eight symbols per thread and ONE aligned store of their 8 (BPSK) / 16 (QPSK) bytes: the row's halfwords come in as two aligned 16-byte loads and pass a funnel
    // shift by the row's (block-uniform) halfword offset H against the output's alignment; the first few symbols of a row, up to the next 8-symbol boundary of the
    // OUTPUT, and its last few go one by one.
    template <int H>
    __device__ __forceinline__ uint4 sd_take8(const uint4 a, const uint4 b)
    {
        const unsigned v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        constexpr int t = H >> 1;
        if constexpr ((H & 1) != 0)
            return uint4{(v[t] >> 16) | (v[t + 1] << 16), (v[t + 1] >> 16) | (v[t + 2] << 16), (v[t + 2] >> 16) | (v[t + 3] << 16), (v[t + 3] >> 16) | (v[t + 4] << 16)};
        else
            return uint4{v[t], v[t + 1], v[t + 2], v[t + 3]};
    }
    template <int H>
    __device__ __forceinline__ void compact8_groups(const short *s0, int groups, int bpsk, int8_t *dst)
    { // s0: the first group's first halfword, H halfwords behind a 16-byte boundary; dst: where the first group's bytes go (8- / 16-byte aligned)
        const uint4 *q = reinterpret_cast<const uint4 *>(s0 - H);
        for (int g = (int)threadIdx.x; g < groups; g += (int)blockDim.x)
        {
            const uint4 a = q[g];
            uint4 b = a;
            if constexpr (H != 0)
                b = q[g + 1];
            const uint4 w = sd_take8<H>(a, b);
            if (bpsk)
            {
                const unsigned lo = (w.x & 0xffu) | ((w.x >> 8) & 0xff00u) | ((w.y & 0xffu) << 16) | ((w.y & 0xff0000u) << 8);
                const unsigned hi = (w.z & 0xffu) | ((w.z >> 8) & 0xff00u) | ((w.w & 0xffu) << 16) | ((w.w & 0xff0000u) << 8);
                *reinterpret_cast<uint2 *>(dst + 8 * (size_t)g) = uint2{lo, hi};
            }
            else
                *reinterpret_cast<uint4 *>(dst + 16 * (size_t)g) = w;
        }
    }
    __global__ __launch_bounds__(256) void k_compact8(const short *sym8, const int *seg, const long long *offsets, int K, int cap, int bpsk, int8_t *soft,
                                                      long long soft_cap)
    {
        const int k = (int)blockIdx.x;
        if (k >= K)
            return;
        const int cnt = seg[2 * k + 1];
        const long long off = offsets[k];
        const short *s = sym8 + (size_t)k * cap + seg[2 * k];
        auto one = [&](int j) {
            const unsigned v = (unsigned short)s[j];
            const long long o = off + j;
            if (bpsk)
            {
                if (o < soft_cap)
                    soft[o] = (int8_t)(v & 0xffu);
            }
            else if (2 * o + 1 < soft_cap)
                *reinterpret_cast<short *>(soft + 2 * o) = (short)v;
        };
        const uintptr_t oaddr = reinterpret_cast<uintptr_t>(soft) + (uintptr_t)(bpsk ? off : 2 * off); // where this row's bytes go
        if (off + cnt > (bpsk ? soft_cap : soft_cap / 2) || (!bpsk && (oaddr & 1) != 0))
        { // (the engine refuses a call whose symbols do not fit before it gets here; an odd output address has no aligned stores)
            for (int j = (int)threadIdx.x; j < cnt; j += (int)blockDim.x)
                one(j);
            return;
        }
        const int gb = bpsk ? 8 : 16, ob = bpsk ? 1 : 2;
        const int to_boundary = (int)(((gb - (oaddr & (uintptr_t)(gb - 1))) & (uintptr_t)(gb - 1)) / ob);
        const int head = to_boundary < cnt ? to_boundary : cnt;
        const int groups = (cnt - head) / 8;
        if ((int)threadIdx.x < head)
            one((int)threadIdx.x);
        const short *s0 = s + head;
        int8_t *dst = soft + (bpsk ? off + head : 2 * (off + head));
        switch ((int)((reinterpret_cast<uintptr_t>(s0) >> 1) & 7))
        {
        case 0: compact8_groups<0>(s0, groups, bpsk, dst); break;
        case 1: compact8_groups<1>(s0, groups, bpsk, dst); break;
        case 2: compact8_groups<2>(s0, groups, bpsk, dst); break;
        case 3: compact8_groups<3>(s0, groups, bpsk, dst); break;
        case 4: compact8_groups<4>(s0, groups, bpsk, dst); break;
        case 5: compact8_groups<5>(s0, groups, bpsk, dst); break;
        case 6: compact8_groups<6>(s0, groups, bpsk, dst); break;
        default: compact8_groups<7>(s0, groups, bpsk, dst); break;
        }
        for (int j = head + 8 * groups + (int)threadIdx.x; j < cnt; j += (int)blockDim.x)
            one(j);
    }
    void launch_compact8(const cf32 *sym_scratch, const int *seg, const long long *offsets, int K, int cap, int bpsk, int8_t *soft, long long soft_cap,
                         hipStream_t st)
    {
        if (K <= 0)
            return;
        ProfScope _ps("k_compact8", st);
        hipLaunchKernelGGL(k_compact8, dim3(K), dim3(256), 0, st, reinterpret_cast<const short *>(sym_scratch), seg, offsets, K, cap, bpsk, soft, soft_cap);
    }
    void launch_quantize(const cf32 *sym_scratch, const int *seg, const long long *offsets, int K, int cap, int bpsk, int8_t *soft, long long soft_cap,
                         float *syms, long long syms_cap, hipStream_t st)
    {
        if (K <= 0)
            return;
        ProfScope _ps("k_quantize", st);
        hipLaunchKernelGGL(k_quantize, dim3(K), dim3(256), 0, st, sym_scratch, seg, offsets, K, cap, bpsk, soft, soft_cap, syms, syms_cap);
    }

    __global__ void k_tail_copy(const cf32 *x, long long n, int cnt, ChunkGeom cg, const int *rot, int order, cf32 *out)
    {
        const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (j >= cnt)
            return;
        const long long i = n - cnt + j; // may be negative: older history
        cf32 v = x[i];
        if (i >= 0 && rot)
            v = rot_apply(v, rot[costas_chunk_of(cg, i)], order);
        out[j] = v;
    }
    // x[i] *= exp(+j rot[chunk of i] unit): the Costas chunks' frames turned back into the stream's (exact for order 2 / 4)
    __global__ __launch_bounds__(256) void k_derotate(cf32 *x, long long n, ChunkGeom cg, const int *rot, int order)
    {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i < n)
            x[i] = rot_apply(x[i], rot[costas_chunk_of(cg, i)], order);
    }
    void launch_derotate(cf32 *x, long long n, const ChunkGeom &cg, const int *rot, int order, hipStream_t st)
    {
        ProfScope _ps("k_derotate", st);
        hipLaunchKernelGGL(k_derotate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, cg, rot, order);
    }
    void launch_tail_copy(const cf32 *x, long long n, int cnt, const ChunkGeom &cg, const int *rot, int order, cf32 *out, hipStream_t st)
    {
        ProfScope _ps("k_tail_copy", st);
        hipLaunchKernelGGL(k_tail_copy, dim3((cnt + 63) / 64), dim3(64), 0, st, x, n, cnt, cg, rot, order, out);
    }
} // namespace sdhip
