// dvbs2_demap.hip -- the PLFRAME -> soft bits stage of the DVB-S2 demodulator on gfx950:
// dvbs2::S2BBToSoft::work (plugins/dvb_support/dvbs2/dvbs2_bb_to_soft.cpp:18-69) on a batch of PL-synchronised, phase-recovered frames
// as S2PLLBlock writes them (dvbs2_pll.cpp:31-49: 90 header symbols, then frame_slot_count slots of 90 symbols):
//   * PLS decode: the 64 header symbols behind the SOF, turned by -pi/4, sliced, compared with the 128 PLS code words over their low 60 bits
//     (dvbs2_bb_to_soft.h:29-41 -- checkSyncMarker starts at bit 59), first minimum wins                                   -> k_s2_pls
//   * PL descrambling: the Gold sequence n = 0 restarted at every frame (s2_scrambling.cpp:10-36), a turn by a multiple of 90 degrees --
//     exact component swaps and negations (s2_scrambling.cpp:45-70)
//   * soft demapping through the constellation's table (constellation_t::demod_soft_lut, constellation.cpp:324-352): index
//     (int)((double)re / 1.5 * res + res / 2) clamped to the table, `bits` int8 per symbol. The TABLE is an input: the caller hands over what
//     constellation_t::make_lut(256) built on the host (module_dvbs2_demod.cpp:123-124) -- its entries come out of expf / logf / hypotf of
//     the host's libm and are data here, not arithmetic to be re-derived. 32APSK has no table in the reference (it evaluates the
//     exponentials per sample): refused.
//   * the block's pilots branch as it is (dvbs2_bb_to_soft.cpp:57-63): every 1476 symbols the write position falls back by 36, the loop still
//     runs over frame_slot_count * 90 input symbols; later symbols overwrite earlier ones, positions behind the last write keep the
//     scratch buffer's previous contents (never initialised in the reference: zero here)                                  -> k_s2_demap
//   * the de-interleaver (s2_deinterleaver.cpp) on the result: sdhip_s2_deinterleave_dev's kernel
// All byte / index work apart from one double division per component: HBM-bound (8 B in, `bits` B out per symbol), a thread per OUTPUT symbol.
#include "../../include/sdhip.h"
#include "common.h"
#include "dvbs2_stages.h"

#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstring>
#include <vector>

namespace sdhip
{
    // ---- host tables
    // PLS code words (ETSI EN 302 307-1 5.5.2.4; dvbs2/s2_defs.h:40-86): (7, 64) code from the 6 x 32 generator, each bit doubled / paired with its
    // complement by the pilots bit, scrambled
    static void s2_pls_codewords(unsigned long long *cw)
    {
        static const unsigned G[6] = {0x55555555u, 0x33333333u, 0x0f0f0f0fu, 0x00ff00ffu, 0x0000ffffu, 0xffffffffu};
        for (int index = 0; index < 128; index++)
        {
            unsigned y = 0;
            for (int row = 0; row < 6; row++)
                if ((index >> (6 - row)) & 1)
                    y ^= G[row];
            unsigned long long code = 0;
            for (int bit = 31; bit >= 0; bit--)
            {
                const unsigned long long yi = (y >> bit) & 1;
                code = (code << 2) | (yi << 1) | ((index & 1) ? (yi ^ 1) : yi);
            }
            cw[index] = code ^ 0x719d83c953422dfaull;
        }
    }
    // PL scrambling sequence Rn (two bits per symbol), Gold code number 0 (s2_scrambling.cpp:10-36): x^18 + x^7 + 1 and x^18 + x^10 + x^7 + x^5 + 1
    static void s2_gold_rn(std::vector<unsigned char> &rn, int count)
    {
        auto lx = [](unsigned X) { const unsigned bit = ((X >> 7) ^ X) & 1u; return ((bit << 18) | X) >> 1; };
        auto ly = [](unsigned Y) { const unsigned bit = ((Y >> 10) ^ (Y >> 7) ^ (Y >> 5) ^ Y) & 1u; return ((bit << 18) | Y) >> 1; };
        std::vector<unsigned char> z(2 * 131072);
        unsigned x = 0x00001u, y = 0x3ffffu;
        for (int i = 0; i < 2 * 131072; i++)
        {
            z[i] = (unsigned char)((x ^ y) & 1u);
            x = lx(x);
            y = ly(y);
        }
        rn.resize(count);
        for (int i = 0; i < count; i++)
            rn[i] = (unsigned char)(z[i] | (z[i + 131072] << 1));
    }
    // get_dvbs2_cfg (codings/dvb-s2/modcod_to_cfg.h:19-151): MODCOD -> modulation (bits per symbol), slots per frame, code rate (dvbs2_code_rate_t)
    S2Cfg s2_cfg_of(int modcod, int shortframes)
    {
        static const int qpsk[11] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11};                  // 1/4 1/3 2/5 1/2 3/5 2/3 3/4 4/5 5/6 8/9 9/10
        static const int psk8[6] = {4, 5, 6, 8, 10, 11}, apsk16[6] = {5, 6, 7, 8, 10, 11}; // 3/5 2/3 3/4 5/6 8/9 9/10 | 2/3 3/4 4/5 5/6 8/9 9/10
        if (modcod >= 1 && modcod < 12)
            return S2Cfg{2, shortframes ? 90 : 360, qpsk[modcod - 1], 0};
        if (modcod >= 12 && modcod < 18)
            return S2Cfg{3, shortframes ? 60 : 240, psk8[modcod - 12], 1};
        if (modcod >= 18 && modcod < 24)
            return S2Cfg{4, shortframes ? 45 : 180, apsk16[modcod - 18], 2};
        if (modcod >= 24 && modcod < 29)
            throw HipError("dvbs2 bb_to_soft: 32APSK has no demapper table in the reference (constellation.cpp:326, 354-357): not on the HIP path");
        throw HipError(modcod <= 0 ? "MODCOD cannot be <= 0!" : "MODCOD not (yet?) supported!"); // modcod_to_cfg.h:26, 146
    }

    // ---- kernels
    __global__ __launch_bounds__(64) void k_s2_pls(const float2 *__restrict__ frames, int frame_stride, int nframes, const unsigned long long *__restrict__ cw, int *pls)
    {
        const int f = (int)blockIdx.x, y = (int)threadIdx.x;
        if (f >= nframes)
            return;
        const float2 v = frames[(size_t)f * frame_stride + 26 + y];
        // input * complex_t(cos(-M_PI / 4), sin(-M_PI / 4)) -- the doubles narrowed to complex_t's floats -- real part: re * c - im * s
        const float c = (float)0.70710678118654757, s = (float)-0.70710678118654746;
        const bool value = (v.x * c - v.y * s) > 0.0f;
        const unsigned long long hdr = __ballot(!value); // lane y -> bit y; the reference shifts bit y = 0 in first: MSB first
        if (y == 0)
        {
            const unsigned long long plheader = ((unsigned long long)__brev((unsigned)hdr) << 32) | (unsigned long long)__brev((unsigned)(hdr >> 32));
            int best = 0, diffs = 64;
            for (int k = 0; k < 128; k++)
            {
                const int d = __popcll((cw[k] ^ plheader) & 0x0FFFFFFFFFFFFFFFull); // checkSyncMarker: bits 59 .. 0
                if (d < diffs)
                {
                    best = k;
                    diffs = d;
                }
            }
            pls[f] = best;
        }
    }

    // one thread per output symbol position j of a frame; nsym = frame_slot_count * 90 input symbols are looked at
    template <int BITS>
    __global__ __launch_bounds__(256) void k_s2_demap(const float2 *__restrict__ frames, int frame_stride, int nframes, int nsym, int pilots,
                                                      const unsigned char *__restrict__ rn, const signed char *__restrict__ lut, int res, signed char *__restrict__ out)
    {
        const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x), f = (int)blockIdx.y;
        if (j >= nsym || f >= nframes)
            return;
        // which input symbol i lands on position j LAST: i - 36 * (i / 1476) = j (pilots), i = j otherwise
        int i = j;
        if (pilots)
        {
            const int s = j / 1440;
            i = j + 36 * s; // segment s writes [1440 s, 1440 s + 1476)
            if (i >= nsym)
            { // ... unless the loop ends first: then the tail of segment s - 1 (its last 36 symbols), if any
                i = (s > 0 && j - 1440 * (s - 1) < 1476) ? j + 36 * (s - 1) : -1;
                if (i >= nsym)
                    i = -1;
            }
        }
        signed char *o = out + ((size_t)f * nsym + j) * BITS;
        if (i < 0)
        { // never written by the reference (scratch buffer contents): zero
#pragma unroll
            for (int b = 0; b < BITS; b++)
                o[b] = 0;
            return;
        }
        const float2 p = frames[(size_t)f * frame_stride + 90 + i];
        float re = p.x, im = p.y;
        switch (rn[i])
        { // S2Scrambling::descramble, s2_scrambling.cpp:45-70
        case 3:
            re = -p.y;
            im = p.x;
            break;
        case 2:
            re = -p.x;
            im = -p.y;
            break;
        case 1:
            re = p.y;
            im = -p.x;
            break;
        default:
            break;
        }
        // int x = (sample.real / 1.5) * lut_resolution + lut_resolution / 2: double arithmetic, truncation, clamp (constellation.cpp:328-341)
        int x = (int)(((double)re / 1.5) * (double)res + (double)(res / 2));
        int y = (int)(((double)im / 1.5) * (double)res + (double)(res / 2));
        x = x < 0 ? 0 : (x >= res ? res - 1 : x);
        y = y < 0 ? 0 : (y >= res ? res - 1 : y);
        const signed char *l = lut + ((size_t)x * res + y) * BITS;
#pragma unroll
        for (int b = 0; b < BITS; b++)
            o[b] = l[b];
    }

    // ---- PL synchroniser: dvbs2::S2PLSyncBlock::work2 (plugins/dvb_support/dvbs2/dvbs2_pl_sync.cpp:52-125) ----------------------------------------
    // One frame's search: for every offset ss of a raw_frame_size window, the differential correlation of the 90 header symbols (conjugate
    // of the previous symbol times the symbol: VOLK's generic conjugate / multiply, the float operations in their order) against the SOF and
    // the PLS scrambling pattern (correlate_sof_diff / correlate_plscode_diff, :127-154), the better of the pilots-off / pilots-on sums scaled
    // by 1 / 57, its magnitude as a double. The block walks ss upwards, keeps the running maximum among the offsets with d.imag > 0 and stops
    // at the first one above `thresold`: = the FIRST qualifying offset above the threshold if there is one (everything in front of it is below
    // it), else the first occurrence of the maximum (> 0), else 0. A block of 256 threads per speculated frame, a thread per offset.
    struct PlsRes
    {
        int best_pos;
    };
    __device__ __forceinline__ void s2_hdr_corr(const float2 *w, double &difference, bool &qual)
    {
        const unsigned dsof = 0x18d2e82u ^ (0x18d2e82u >> 1);
        const unsigned long long dscr = 0x719d83c953422dfaull ^ (0x719d83c953422dfaull >> 1);
        float sr = 0.0f, si = 0.0f, pr = 0.0f, pi = 0.0f; // csof, cplsc
        float2 prev = w[0];
        // plheader_symbols[0] = 0 * corr[ss]: adds (or subtracts) a zero to csof -- nothing
        for (int i = 1; i < 90; i++)
        {
            const float2 cur = w[i];
            // conj(prev) * cur: (a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re) with a = (prev.x, -prev.y)
            const float nre = -prev.y;
            const float dre = prev.x * cur.x - nre * cur.y;
            const float dim = prev.x * cur.y + nre * cur.x;
            if (i < 26)
            {
                if (((dsof >> (25 - i)) ^ (unsigned)i) & 1u)
                {
                    sr += dre;
                    si += dim;
                }
                else
                {
                    sr -= dre;
                    si -= dim;
                }
            }
            else
            {
                const int k = i - 26; // diffs index within the PLS field: odd ones only
                if (k & 1)
                {
                    if ((dscr >> (63 - k)) & 1ull)
                    {
                        pr -= dre;
                        pi -= dim;
                    }
                    else
                    {
                        pr += dre;
                        pi += dim;
                    }
                }
            }
            prev = cur;
        }
        const float c0r = sr + pr, c0i = si + pi, c1r = sr - pr, c1i = si - pi;
        const float n0 = sqrtf(c0r * c0r + c0i * c0i), n1 = sqrtf(c1r * c1r + c1i * c1i);
        const float cr = n0 > n1 ? c0r : c1r, ci = n0 > n1 ? c0i : c1i;
        const float sc = 1.0f / (float)(26 - 1 + 64 / 2);
        const float dr = cr * sc, di = ci * sc;
        difference = (double)sqrtf(dr * dr + di * di);
        qual = di > 0.0f;
    }
    __global__ __launch_bounds__(256) void k_s2_plsync_search(const float2 *__restrict__ syms, long long base, int raw, int nframes, float thresold, int *best_pos)
    {
        const int f = (int)blockIdx.x;
        if (f >= nframes)
            return;
        const float2 *win = syms + base + (long long)f * raw;
        const int nss = raw - 90;
        int first_over = 0x7fffffff, arg = 0x7fffffff;
        double best = 0.0;
        for (int ss = (int)threadIdx.x; ss < nss; ss += 256)
        {
            double d;
            bool q;
            s2_hdr_corr(win + ss, d, q);
            if (!q)
                continue;
            if (d > (double)thresold && ss < first_over)
                first_over = ss;
            if (d > best) // this thread's offsets ascend: strict > keeps the first occurrence
            {
                best = d;
                arg = ss;
            }
        }
        __shared__ int s_first[256], s_arg[256];
        __shared__ double s_best[256];
        s_first[threadIdx.x] = first_over;
        s_arg[threadIdx.x] = arg;
        s_best[threadIdx.x] = best;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1)
        {
            if ((int)threadIdx.x < st)
            {
                const int o = (int)threadIdx.x + st;
                if (s_first[o] < s_first[threadIdx.x])
                    s_first[threadIdx.x] = s_first[o];
                if (s_best[o] > s_best[threadIdx.x] || (s_best[o] == s_best[threadIdx.x] && s_arg[o] < s_arg[threadIdx.x]))
                {
                    s_best[threadIdx.x] = s_best[o];
                    s_arg[threadIdx.x] = s_arg[o];
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
            best_pos[f] = s_first[0] != 0x7fffffff ? s_first[0] : (s_best[0] > 0.0 ? s_arg[0] : 0);
    }
    // frame k of the output = raw symbols from starts[k]
    __global__ __launch_bounds__(256) void k_s2_plsync_emit(const float2 *__restrict__ syms, const long long *__restrict__ starts, int raw, int nframes, float2 *out, int stride)
    {
        const int f = (int)blockIdx.y, i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (f < nframes && i < raw)
            out[(size_t)f * stride + i] = syms[starts[f] + i];
    }

    // ---- frame PLL: dvbs2::S2PLLBlock::work (plugins/dvb_support/dvbs2/dvbs2_pll.cpp:20-66) -----------------------------------------------------
    // A second-order loop over EVERY symbol of the synchronised frames, its state carried from frame to frame: rotate by (cosf(-phase),
    // sinf(-phase)); phase error = arg(symbol * conj(known symbol)) on the 90 header symbols (SOF, then the PLS code word of the configured
    // MODCOD), the demapper table's phase_error entry on everything behind; header symbols leave as "proper 45 degree BPSK". One sequential
    // lane, every float operation where the reference has it: glibc 2.35's sinf / cosf (the evaluation demod_kernels.hip carries, copied
    // below) and its atan2f / atanf (fdlibm's float code -- this restatement equals the host libm's results on 6e7 random arguments,
    // special values included: tests/test_dvbs2_pll_math_cpu.py). With pilots the block counts ONE pilot block (update(), dvbs2_pll.h:33-47:
    // it divides the slot count by 90 where the symbol count is meant) and leaves the rest of the frame untouched: so does this.
    __device__ __forceinline__ unsigned s2_abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
    __device__ __forceinline__ void s2_sincosf(float y, float &sn, float &cs)
    { // = sdhip::sd_sincosf (demod_kernels.hip): sinf(y), cosf(y) of glibc, one reduction
        const double x0 = (double)y;
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double r = x0 * hpi_inv;
        const int n = ((int)r + 0x800000) >> 24;
        const double xr = fma(-(double)n, hpi, x0);
        const double xs = ((n + 1) & 2) ? -xr : xr;
        const double x2 = xr * xr;
        const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
        const double x3 = xs * x2;
        const double s1 = fma(x2, s3c, s2c);
        const double x7 = x3 * x2;
        const double sp = fma(x7, s1, fma(x3, s1c, xs));
        const double c1c = -0x1.ffffffd0c621cp-2, c2c = 0x1.55553e1068f19p-5, c3c = -0x1.6c087e89a359dp-10, c4c = 0x1.99343027bf8c3p-16;
        const double x4 = x2 * x2;
        const double c2 = fma(x2, c4c, c3c);
        const double c1 = fma(x2, c1c, 0x1p0);
        const double x6 = x4 * x2;
        const double cp0 = fma(x6, c2, fma(x4, c2c, c1));
        const float fs = (float)sp, fc0 = (float)cp0;
        const float fc = (n & 2) ? -fc0 : fc0;
        const bool odd = (n & 1) != 0;
        const bool tiny = s2_abstop12(y) < s2_abstop12(0x1p-12f);
        sn = tiny ? y : (odd ? fc : fs);
        cs = tiny ? 1.0f : (odd ? fs : fc);
    }
    __device__ __forceinline__ float s2_atanf(float x)
    { // glibc 2.35 sysdeps/ieee754/flt-32/s_atanf.c (fdlibm): argument reduction to one of four intervals, odd polynomial of degree 11 in x^2
        const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
        const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
        const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                              6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
        const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff;
        int id;
        if (ix >= 0x4c000000)
        {
            if (ix > 0x7f800000)
                return x + x;
            return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
        }
        if (ix < 0x3ee00000)
        {
            if (ix < 0x31000000)
                return x; // (huge + x > one: raises inexact, returns x)
            id = -1;
        }
        else
        {
            x = fabsf(x);
            if (ix < 0x3f980000)
            {
                if (ix < 0x3f300000)
                {
                    id = 0;
                    x = (2.0f * x - 1.0f) / (2.0f + x);
                }
                else
                {
                    id = 1;
                    x = (x - 1.0f) / (x + 1.0f);
                }
            }
            else if (ix < 0x401c0000)
            {
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            }
            else
            {
                id = 3;
                x = -1.0f / x;
            }
        }
        const float z = x * x, w = z * z;
        const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
        const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
        if (id < 0)
            return x - x * (s1 + s2);
        const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
        return hx < 0 ? -r : r;
    }
    __device__ __forceinline__ float s2_atan2f(float y, float x)
    { // glibc 2.35 sysdeps/ieee754/flt-32/e_atan2f.c (fdlibm)
        const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
        const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff, hy = (int)__float_as_uint(y), iy = hy & 0x7fffffff;
        if (ix > 0x7f800000 || iy > 0x7f800000)
            return x + y;
        if (hx == 0x3f800000)
            return s2_atanf(y);
        const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
        if (iy == 0)
            return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
        if (ix == 0)
            return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
        if (ix == 0x7f800000)
        {
            if (iy == 0x7f800000)
                return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
            return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
        }
        if (iy == 0x7f800000)
            return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
        const int k = (iy - ix) >> 23;
        float z;
        if (k > 60)
            z = pi_o_2 + 0.5f * pi_lo;
        else if (hx < 0 && k < -60)
            z = 0.0f;
        else
            z = s2_atanf(fabsf(y / x));
        if (m == 0)
            return z;
        if (m == 1)
            return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
        if (m == 2)
            return pi - (z - pi_lo);
        return (z - pi_lo) - pi;
    }
    // unit test hook of the two functions above (host twin and GPU): out[i] = atan2f(y[i], x[i])
    __global__ void k_s2_atan2f(const float *y, const float *x, int n, float *out)
    {
        const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (i < n)
            out[i] = s2_atan2f(y[i], x[i]);
    }
    struct S2PllCtx
    {
        float alpha, beta;
        const float2 *hdr;    // 90 known header symbols
        const float *lut_err; // the demapper table's phase errors
        int res;
    };
    // one step of S2PLLBlock::work's loop (dvbs2_pll.cpp:31-62) on symbol i of a frame: every float operation where the reference has it
    __device__ __forceinline__ float2 s2_pll_step(const S2PllCtx &c, int i, const float2 v, float &phase, float &freq)
    {
        float sn, cs;
        s2_sincosf(-phase, sn, cs);
        const float tr = (v.x * cs) - (v.y * sn);
        const float ti = (v.y * cs) + (v.x * sn);
        float error;
        float2 o;
        if (i >= 90)
        { // constellation->demod_soft_lut(tmp_val, nullptr, &error), constellation.cpp:324-352
            int ix = (int)(((double)tr / 1.5) * (double)c.res + (double)(c.res / 2));
            int iy = (int)(((double)ti / 1.5) * (double)c.res + (double)(c.res / 2));
            ix = ix < 0 ? 0 : (ix >= c.res ? c.res - 1 : ix);
            iy = iy < 0 ? 0 : (iy >= c.res ? c.res - 1 : iy);
            error = c.lut_err[(size_t)ix * c.res + iy];
            o = make_float2(tr, ti);
        }
        else
        { // (tmp_val * known.conj()).arg(): (a.re * b.re - a.im * b.im, a.im * b.re + a.re * b.im) with b = (k.x, -k.y)
            const float2 k = c.hdr[i];
            const float nb = -k.y;
            const float pr = (tr * k.x) - (ti * nb);
            const float pim = (ti * k.x) + (tr * nb);
            error = s2_atan2f(pim, pr);
            o = (i & 1) ? make_float2(-tr, ti) : make_float2(ti, tr);
        }
        freq = freq + c.beta * error;
        phase = phase + (freq + c.alpha * error);
        // while (phase > 2 pi) phase -= 2 pi; while (phase < -2 pi) phase += 2 pi: float compared with the double constant, the step in double
        while ((double)phase > 2.0 * 3.14159265358979323846)
            phase = (float)((double)phase - 2.0 * 3.14159265358979323846);
        while ((double)phase < -2.0 * 3.14159265358979323846)
            phase = (float)((double)phase + 2.0 * 3.14159265358979323846);
        if (freq > 1.0f)
            freq = 1.0f;
        if (freq < -1.0f)
            freq = -1.0f;
        return o;
    }
    // trace (may be null): the loop frequency every 1024 steps (what a new stream's acquisition leaves for the estimates' branch, S2Pll::run)
    __global__ __launch_bounds__(64) void k_s2_pll_seq(const float2 *__restrict__ in, float2 *__restrict__ out, int stride, int nframes, int per_frame, S2PllCtx c, S2PllState *state,
                                                       float *__restrict__ trace)
    {
        if (blockIdx.x != 0 || threadIdx.x != 0)
            return;
        float phase = state->phase, freq = state->freq;
        long long g = 0;
        for (int f = 0; f < nframes; f++)
        {
            const float2 *x = in + (size_t)f * stride;
            float2 *o = out + (size_t)f * stride;
            for (int i = 0; i < per_frame; i++, g++)
            {
                o[i] = s2_pll_step(c, i, x[i], phase, freq);
                if (trace && (g & 1023) == 1023)
                    trace[g >> 10] = freq;
            }
        }
        state->phase = phase;
        state->freq = freq;
    }

    // ---- the frame-parallel schedule of the same loop (round 4) ---------------------------------------------------------------------
    // The loop's steps over a batch of frames form ONE chain g = f * per_frame + i. It is cut into lanes of L steps; a lane starts W steps in
    // front of its range (on the previous frame's tail where it has to) from a DATA-AIDED estimate -- every frame begins with 90 known symbols:
    // z_f = sum conj(known) * received is the loop's phase at the header's centre, consecutive headers give the frequency (the carried loop
    // frequency only picks the 2 pi / per_frame branch) -- walks the warm-up without storing, leaves the state it reaches its range with
    // (`start`), then its range with stores, and leaves its end state. Lane 0 continues the carried state exactly. The host certifies the
    // chain: start[l] against end[l - 1] (phase modulo the turn, frequency), re-runs the lanes that miss from their predecessor's exact end state.
    // What this can promise is NOT symbols within 1e-5 of the serial loop's: the detector is a 256 x 256 table (piecewise-constant feedback), two
    // trajectories of the loop on the same symbols hover ~1e-2 (8PSK, 10 dB) ... 2e-4 (QPSK, 8 dB) of a symbol apart for good
    // (tools/s2_pll_frame_study.py) -- the contract is the decoders' output: the same BBFRAMEs.
    // z[f]: the header of frame f against the known one (the loop's phase at the header's centre). a[f] (may be null): the wiped header against itself 45 symbols
    // on, sum_{i >= 45} w_i conj(w_{i-45}), w_i = received_i conj(known_i): its angle / 45 is the carrier's rate INSIDE the header -- unambiguous to +-pi / 45 per
    // symbol, coarse per frame (sigma ~5e-3 rad / symbol at 10 dB), and over the frames of a long call good to a few 1e-5: what picks the 2 pi / per_frame branch of
    // the header-to-header estimate when the loop's own frequency cannot (a new stream with a carrier offset the loop has not pulled in yet)
    __global__ __launch_bounds__(64) void k_s2_hdr_est(const float2 *__restrict__ in, int stride, int nframes, const float2 *__restrict__ hdr, double2 *__restrict__ z,
                                                       double2 *__restrict__ a)
    {
        __shared__ float2 w[90];
        const int f = (int)blockIdx.x, t = (int)threadIdx.x;
        if (f >= nframes)
            return;
        double re = 0.0, im = 0.0;
        for (int i = t; i < 90; i += 64)
        {
            const float2 v = in[(size_t)f * stride + i], k = hdr[i];
            const float wr = v.x * k.x + v.y * k.y, wi = v.y * k.x - v.x * k.y; // received * conj(known)
            w[i] = make_float2(wr, wi);
            re += (double)v.x * k.x + (double)v.y * k.y;
            im += (double)v.y * k.x - (double)v.x * k.y;
        }
        __syncthreads();
        double ar = 0.0, ai = 0.0;
        if (a && t < 45)
        {
            const float2 p = w[t + 45], q = w[t];
            ar = (double)p.x * q.x + (double)p.y * q.y;
            ai = (double)p.y * q.x - (double)p.x * q.y;
        }
        for (int o = 32; o > 0; o >>= 1)
        {
            re += __shfl_xor(re, o);
            im += __shfl_xor(im, o);
            ar += __shfl_xor(ar, o);
            ai += __shfl_xor(ai, o);
        }
        if (t == 0)
        {
            z[f] = make_double2(re, im);
            if (a)
                a[f] = make_double2(ar, ai);
        }
    }
    struct S2PllLane
    {
        long long g0, g1;  // the lane's range of chain steps
        int warm;          // steps walked in front of g0 (0: the lane continues `from` exactly)
        int from;          // >= 0: start from end[from]; -1: from the carried state; -2: from the estimate
    };
    __global__ __launch_bounds__(64) void k_s2_pll_lanes(const float2 *__restrict__ in, float2 *__restrict__ out, int stride, int per_frame, S2PllCtx c,
                                                         const S2PllLane *__restrict__ lanes, const int *__restrict__ list, int nlist, const double2 *__restrict__ z, int nframes,
                                                         float freq_hint, const S2PllState *__restrict__ carried, S2PllState *__restrict__ start, S2PllState *__restrict__ end)
    {
        const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (k >= nlist)
            return;
        const int l = list ? list[k] : k;
        const S2PllLane ln = lanes[l];
        float phase, freq;
        long long g = ln.g0 - ln.warm;
        if (ln.from >= 0)
        {
            phase = end[ln.from].phase;
            freq = end[ln.from].freq;
        }
        else if (ln.from == -1)
        {
            phase = carried->phase;
            freq = carried->freq;
        }
        else
        { // estimate at step g from the header of the frame g0 lies in (and its neighbour's)
            const int F = (int)(ln.g0 / per_frame);
            const int A = F > 0 ? F - 1 : 0, B = F > 0 ? F : (nframes > 1 ? 1 : 0);
            double fr = (double)freq_hint;
            if (B != A)
            {
                const double2 za = z[A], zb = z[B];
                const double d = atan2(zb.y * za.x - zb.x * za.y, zb.x * za.x + zb.y * za.y); // arg(zb * conj(za))
                const double turns = rint((fr * (double)per_frame - d) / (2.0 * 3.14159265358979323846));
                fr = (d + 2.0 * 3.14159265358979323846 * turns) / (double)per_frame;
            }
            const double2 zf = z[F];
            double ph = atan2(zf.y, zf.x) + fr * ((double)(g - (long long)F * per_frame) - 44.5);
            ph -= 2.0 * 3.14159265358979323846 * rint(ph / (2.0 * 3.14159265358979323846));
            phase = (float)ph;
            freq = (float)fr;
        }
        int f = (int)(g / per_frame), i = (int)(g - (long long)f * per_frame);
        const float2 *x = in + (size_t)f * stride;
        for (; g < ln.g0; g++)
        { // warm-up: the loop runs, nothing is stored
            (void)s2_pll_step(c, i, x[i], phase, freq);
            if (++i == per_frame)
            {
                i = 0;
                f++;
                x = in + (size_t)f * stride;
            }
        }
        start[l].phase = phase;
        start[l].freq = freq;
        float2 *o = out + (size_t)f * stride;
        for (; g < ln.g1; g++)
        {
            o[i] = s2_pll_step(c, i, x[i], phase, freq);
            if (++i == per_frame)
            {
                i = 0;
                f++;
                x = in + (size_t)f * stride;
                o = out + (size_t)f * stride;
            }
        }
        end[l].phase = phase;
        end[l].freq = freq;
    }

    // ---- host side of the frame PLL ----------------------------------------------------------------------------------------------------------
    int s2_raw_frame_size(int slot_number, int pilots)
    { // S2PLSyncBlock's constructor, dvbs2_pl_sync.cpp:12-30
        int raw = (slot_number + 1) * 90;
        if (pilots)
        {
            int raw_size = (raw - 90) / 90, pilot_cnt = 1;
            raw_size -= 16;
            while (raw_size > 16)
            {
                raw_size -= 16;
                pilot_cnt++;
            }
            raw += pilot_cnt * 36;
        }
        return raw;
    }
    int s2_pll_walked(int slots, int pilots)
    { // S2PLLBlock::update(), dvbs2_pll.h:33-47 (frame_slot_count is the SLOT count: the loop below it never runs, one pilot block is counted)
        int pilot_cnt = 0;
        if (pilots)
        {
            int raw_size = (slots - 90) / 90;
            pilot_cnt = 1;
            raw_size -= 16;
            while (raw_size > 16)
            {
                raw_size -= 16;
                pilot_cnt++;
            }
        }
        return (slots + 1) * 90 + pilot_cnt * 36;
    }
    static long env_long(const char *name, long dflt)
    {
        const char *v = getenv(name);
        return (v && *v) ? atol(v) : dflt;
    }
    struct S2PllImpl
    {
        int device = 0, per_frame = 0;
        S2PllCtx ctx{};
        DevBuf<float2> d_hdr;
        DevBuf<float> d_lut;
        DevBuf<S2PllState> d_state, d_start, d_end;
        DevBuf<double2> d_z, d_a;
        DevBuf<S2PllLane> d_lanes;
        DevBuf<int> d_list;
        DevBuf<float> d_trace;
        double freq_hint = 0.0; // the loop frequency the estimates pick their branch with: a MEAN (the loop's own frequency state wanders by ~1e-4 rad / symbol
                                // at 10 dB, half the 2 pi / 21 690 branch spacing of normal 8PSK frames)
        std::vector<S2PllLane> lanes;
        std::vector<S2PllState> h_start, h_end;
        std::vector<int> list;
    };
    S2Pll::S2Pll(int device, int modcod, int shortframes, int pilots, float loop_bw, const float *lut_phase_error, int lut_resolution, bool exact_) : exact(exact_), im(new S2PllImpl)
    {
        const S2Cfg c = s2_cfg_of(modcod, shortframes ? 1 : 0);
        if (!lut_phase_error || lut_resolution < 2 || lut_resolution > 4096)
            throw HipError("dvbs2 pll: the demapper table's phase errors and the loop state must be handed over");
        state = S2PllState{0.0f, 0.0f};
        im->device = device;
        im->per_frame = s2_pll_walked(c.slots, pilots);
        SD_HIP(hipSetDevice(device));
        // loop gains, dvbs2_pll.cpp:8-12 (the Costas block's expression)
        const float damping = sqrtf(2.0f) / 2.0f;
        const float denom = (float)(1.0 + 2.0 * damping * loop_bw + loop_bw * loop_bw);
        im->ctx.alpha = (4 * damping * loop_bw) / denom;
        im->ctx.beta = (4 * loop_bw * loop_bw) / denom;
        // known header: s2_sof / s2_plscodes symbols (dvbs2/s2_defs.h:16-36, 74-80), computed with the host's cosf / sinf / sqrtf like the reference's tables
        float2 hdr[90];
        for (int s = 0; s < 26; s++)
        {
            const bool bit = (0x18d2e82u >> (25 - s)) & 1u;
            const int angle = bit * 2 + (s & 1);
            hdr[s].x = 1 * cosf(M_PI / 4 + 2 * M_PI * angle / 4);
            hdr[s].y = 1 * sinf(M_PI / 4 + 2 * M_PI * angle / 4);
        }
        unsigned long long cw[128];
        s2_pls_codewords(cw);
        const unsigned long long code = cw[(modcod << 2) | ((shortframes ? 1 : 0) << 1) | (pilots ? 1 : 0)];
        for (int i = 0; i < 64; i++)
        {
            const int yi = (int)((code >> (63 - i)) & 1ull), nyi = yi ^ (i & 1);
            hdr[26 + i].x = 1 * (1 - 2 * nyi) / sqrtf(2);
            hdr[26 + i].y = 1 * (1 - 2 * yi) / sqrtf(2);
        }
        im->d_hdr.reserve(90);
        im->d_lut.reserve((size_t)lut_resolution * lut_resolution);
        im->d_state.reserve(1);
        SD_HIP(hipMemcpy(im->d_hdr.p, hdr, sizeof(hdr), hipMemcpyHostToDevice));
        SD_HIP(hipMemcpy(im->d_lut.p, lut_phase_error, (size_t)lut_resolution * lut_resolution * sizeof(float), hipMemcpyHostToDevice));
        im->ctx.hdr = im->d_hdr.p;
        im->ctx.lut_err = im->d_lut.p;
        im->ctx.res = lut_resolution;
    }
    S2Pll::~S2Pll() = default;
    int S2Pll::per_frame() const { return im->per_frame; }
    void S2Pll::set_hint(double f) { im->freq_hint = f; }
    void S2Pll::add_frequency(float df)
    {
        state.freq += df;
        im->freq_hint += (double)df;
    }
    static bool s2_state_close(const S2PllState &a, const S2PllState &b, double tol_p, double tol_f)
    {
        double d = (double)a.phase - (double)b.phase;
        d -= 2.0 * M_PI * rint(d / (2.0 * M_PI));
        return fabs(d) < tol_p && fabs((double)a.freq - (double)b.freq) < tol_f;
    }
    void S2Pll::run(const float *d_in, float *d_out, int stride, int nframes, hipStream_t st)
    {
        S2PllImpl &m = *im;
        stats = S2PllStats{};
        if (nframes <= 0)
            return;
        if (stride < m.per_frame)
            throw HipError("dvbs2 pll: frame_stride shorter than the symbols the loop walks");
        SD_HIP(hipSetDevice(m.device));
        const float2 *in = reinterpret_cast<const float2 *>(d_in);
        float2 *out = reinterpret_cast<float2 *>(d_out);
        auto serial = [&](int f0, int nf, bool trace)
        {
            SD_HIP(hipMemcpyAsync(m.d_state.p, &state, sizeof(state), hipMemcpyHostToDevice, st));
            const size_t ntr = trace ? (size_t)(((long long)nf * m.per_frame) >> 10) : 0;
            if (ntr)
                m.d_trace.reserve(ntr);
            {
                ProfScope _ps("k_s2_pll_seq", st);
                hipLaunchKernelGGL(k_s2_pll_seq, dim3(1), dim3(64), 0, st, in + (size_t)f0 * stride, out + (size_t)f0 * stride, stride, nf, m.per_frame, m.ctx, m.d_state.p,
                                   ntr ? m.d_trace.p : (float *)nullptr);
            }
            SD_HIP(hipMemcpyAsync(&state, m.d_state.p, sizeof(state), hipMemcpyDeviceToHost, st));
            SD_HIP(hipStreamSynchronize(st));
            stats.serial_frames += (unsigned)nf;
            if (ntr)
            { // mean loop frequency over the second half of the stretch
                std::vector<float> tr(ntr);
                SD_HIP(hipMemcpy(tr.data(), m.d_trace.p, ntr * sizeof(float), hipMemcpyDeviceToHost));
                double a = 0.0;
                const size_t h0 = ntr / 2;
                for (size_t i = h0; i < ntr; i++)
                    a += tr[i];
                m.freq_hint = a / (double)(ntr - h0);
            }
            else
                m.freq_hint = state.freq;
        };
        if (exact)
        {
            serial(0, nframes, false);
            return;
        }
        int f0 = 0;
        if (!have_hint)
        { // a new stream: the loop acquires on its own (the estimates need its frequency to pick their 2 pi / per_frame branch)
            const long acq_syms = env_long("SDHIP_S2PLL_ACQ", 65536);
            const int acq = (int)std::min<long>(nframes, std::max<long>(2, (acq_syms + m.per_frame - 1) / m.per_frame));
            serial(0, acq, true);
            f0 = acq;
            have_hint = true;
            if (f0 >= nframes)
                return;
        }
        const int nf = nframes - f0;
        in += (size_t)f0 * stride;
        out += (size_t)f0 * stride;
        const long long total = (long long)nf * m.per_frame;
        const long W = std::max<long>(0, env_long("SDHIP_S2PLL_W", 2048));
        long L = env_long("SDHIP_S2PLL_L", 0);
        if (L <= 0) // up to 16 384 lanes (256 waves: a wave per CU) of at least 2048 steps: the lanes' serial walk is the stage's time, re-run rounds cost one more each
            L = std::max<long long>(2048, (total + 16383) / 16384);
        L = std::max<long>(L, std::max<long>(W, 64));
        const double tol_p = (double)env_long("SDHIP_S2PLL_TOL_MRAD", 100) * 1e-3, tol_f = (double)env_long("SDHIP_S2PLL_TOL_UFREQ", 250) * 1e-6;
        const int max_rounds = (int)env_long("SDHIP_S2PLL_ROUNDS", 6);
        m.lanes.clear();
        for (long long g = 0; g < total; g += L)
        {
            S2PllLane ln;
            ln.g0 = g;
            ln.g1 = std::min<long long>(total, g + L);
            ln.warm = g == 0 ? 0 : (int)std::min<long long>(W, g);
            ln.from = g == 0 ? -1 : -2;
            m.lanes.push_back(ln);
        }
        const int nl = (int)m.lanes.size();
        stats.lanes = (unsigned)nl;
        m.d_lanes.reserve(nl);
        m.d_start.reserve(nl);
        m.d_end.reserve(nl);
        m.d_z.reserve(nf);
        m.d_list.reserve(nl);
        SD_HIP(hipMemcpyAsync(m.d_lanes.p, m.lanes.data(), nl * sizeof(S2PllLane), hipMemcpyHostToDevice, st));
        SD_HIP(hipMemcpyAsync(m.d_state.p, &state, sizeof(state), hipMemcpyHostToDevice, st));
        {
            ProfScope _ps("k_s2_hdr_est", st);
            m.d_a.reserve(nf);
            hipLaunchKernelGGL(k_s2_hdr_est, dim3((unsigned)nf), dim3(64), 0, st, in, stride, nf, m.ctx.hdr, m.d_z.p, m.d_a.p);
        }
        m.h_start.resize(nl);
        m.h_end.resize(nl);
        auto fetch = [&]()
        {
            SD_HIP(hipMemcpyAsync(m.h_start.data(), m.d_start.p, nl * sizeof(S2PllState), hipMemcpyDeviceToHost, st));
            SD_HIP(hipMemcpyAsync(m.h_end.data(), m.d_end.p, nl * sizeof(S2PllState), hipMemcpyDeviceToHost, st));
            SD_HIP(hipStreamSynchronize(st));
        };
        auto count_bad = [&]()
        {
            int nb = 0;
            for (int l = 1; l < nl; l++)
                nb += !s2_state_close(m.h_start[l], m.h_end[l - 1], tol_p, tol_f);
            return nb;
        };
        auto launch_all = [&](double hint)
        {
            ProfScope _ps("k_s2_pll_lanes", st);
            hipLaunchKernelGGL(k_s2_pll_lanes, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, st, in, out, stride, m.per_frame, m.ctx, m.d_lanes.p, (const int *)nullptr, nl, m.d_z.p, nf,
                               (float)hint, m.d_state.p, m.d_start.p, m.d_end.p);
        };
        launch_all(m.freq_hint);
        fetch();
        // The estimates' frequency comes from two headers modulo 2 pi / per_frame; the hint picks the branch. When a quarter of the chain misses, the
        // hint was on the wrong side of a branch boundary (a new stream whose acquisition stretch was short, a frequency step): try its neighbours
        // and keep the branch the chain agrees with.
        if (nl >= 8 && count_bad() > nl / 4)
        {
            const double sp = 2.0 * M_PI / (double)m.per_frame;
            const int bad0 = count_bad();
            int best = bad0;
            double best_hint = m.freq_hint;
            double centre = m.freq_hint;
            if (nf >= 128)
            { // enough headers for the rate INSIDE them to name the branch (k_s2_hdr_est's second output): search around that instead of around the loop's word
                std::vector<double> ha(2 * (size_t)nf);
                SD_HIP(hipMemcpyAsync(ha.data(), m.d_a.p, ha.size() * sizeof(double), hipMemcpyDeviceToHost, st));
                SD_HIP(hipStreamSynchronize(st));
                double sr = 0.0, si = 0.0;
                for (int f = 0; f < nf; f++)
                {
                    sr += ha[2 * (size_t)f];
                    si += ha[2 * (size_t)f + 1];
                }
                centre = atan2(si, sr) / 45.0;
                if (getenv("SDHIP_DEBUG"))
                    fprintf(stderr, "[sdhip] s2 pll: rate inside %d headers %.3e rad/symbol (the loop's hint was %.3e)\n", nf, centre, m.freq_hint);
                launch_all(centre);
                fetch();
                stats.branch_tries++;
                const int nb = count_bad();
                if (nb < best)
                {
                    best = nb;
                    best_hint = centre;
                }
            }
            // (round 5: out to +-8 branches, nearest first -- a recording with a carrier offset of a few 1e-4 rad / symbol, which the reference's loop takes a
            // hundred frames to pull in, leaves the two-frame acquisition stretch with a hint several branches off; each try is one launch, once per stream)
            const int reach = (int)env_long("SDHIP_S2PLL_BRANCHES", 8);
            std::vector<int> order;
            for (int a = 1; a <= reach; a++)
            {
                order.push_back(+a);
                order.push_back(-a);
            }
            for (int j : order)
            {
                if (best <= nl / 16)
                    break;
                launch_all(centre + j * sp);
                fetch();
                const int nb = count_bad();
                stats.branch_tries++;
                if (nb < best)
                {
                    best = nb;
                    best_hint = centre + j * sp;
                }
                if (nb <= nl / 16)
                    break;
            }
            if (getenv("SDHIP_DEBUG"))
                fprintf(stderr, "[sdhip] s2 pll: %d of %d lanes missed with the hint %.3e rad/symbol; branch search -> %.3e (%d miss)\n", bad0, nl, m.freq_hint, best_hint, best);
            m.freq_hint = best_hint;
            launch_all(m.freq_hint); // (the last candidate tried is not always the one kept)
            fetch();
        }
        // certify the chain: a lane's start state (behind its warm-up) against its predecessor's end state. Lanes that miss are re-run from the
        // predecessor's EXACT end state -- only the heads of runs of missing lanes can be (their predecessor is settled); after max_rounds the rest is
        // re-run once from whatever their predecessors ended with and let through (a stream the loop is not locked on: noise)
        for (int round = 0;; round++)
        {
            m.list.clear();
            std::vector<char> bad(nl, 0);
            for (int l = 1; l < nl; l++)
                bad[l] = !s2_state_close(m.h_start[l], m.h_end[l - 1], tol_p, tol_f);
            if (getenv("SDHIP_DEBUG"))
                for (int l = 1; l < nl; l++)
                    if (bad[l])
                    {
                        double d = (double)m.h_start[l].phase - (double)m.h_end[l - 1].phase;
                        d -= 2.0 * M_PI * rint(d / (2.0 * M_PI));
                        fprintf(stderr, "[sdhip] s2 pll round %d lane %d (step %lld): start - predecessor's end = %+.4f rad, %+.3e rad/symbol (freq %.3e)\n", round, l, m.lanes[l].g0, d,
                                (double)m.h_start[l].freq - (double)m.h_end[l - 1].freq, (double)m.h_end[l - 1].freq);
                    }
            const bool last = round >= max_rounds;
            for (int l = 1; l < nl; l++)
                if (bad[l] && (last || !bad[l - 1]))
                    m.list.push_back(l);
            if (m.list.empty())
                break;
            for (int l : m.list)
            {
                m.lanes[l].warm = 0;
                m.lanes[l].from = l - 1;
            }
            SD_HIP(hipMemcpyAsync(m.d_lanes.p, m.lanes.data(), nl * sizeof(S2PllLane), hipMemcpyHostToDevice, st));
            SD_HIP(hipMemcpyAsync(m.d_list.p, m.list.data(), m.list.size() * sizeof(int), hipMemcpyHostToDevice, st));
            {
                ProfScope _ps("k_s2_pll_lanes", st);
                hipLaunchKernelGGL(k_s2_pll_lanes, dim3((unsigned)((m.list.size() + 63) / 64)), dim3(64), 0, st, in, out, stride, m.per_frame, m.ctx, m.d_lanes.p, m.d_list.p,
                                   (int)m.list.size(), m.d_z.p, nf, (float)m.freq_hint, m.d_state.p, m.d_start.p, m.d_end.p);
            }
            if (last)
            {
                stats.forced += (unsigned)m.list.size();
                fetch();
                break;
            }
            stats.rerun += (unsigned)m.list.size();
            fetch();
        }
        state = m.h_end[nl - 1];
        // the next call's hint: the mean loop frequency over the lanes' end states
        double a = 0.0;
        for (int l = 0; l < nl; l++)
            a += m.h_end[l].freq;
        m.freq_hint = a / (double)nl;
    }

    struct S2DemapCache
    {
        DevBuf<unsigned long long> d_cw;
        DevBuf<unsigned char> d_rn;
        DevBuf<signed char> d_lut, d_slots;
        int rn_count = 0;
        std::vector<signed char> lut_host;
        int device = -1;
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
    int sdhip_s2_bb_to_soft_dev(int device, int modcod, int shortframes, int pilots, const float *d_plframes, int frame_stride, int nframes, const int8_t *lut_bits,
                                int lut_resolution, int8_t *d_soft, int *d_pls)
    {
        SD_GUARD_BEGIN
        const S2Cfg c = s2_cfg_of(modcod, shortframes ? 1 : 0);
        const int nsym = c.slots * 90;
        if (frame_stride < 90 + nsym)
            throw HipError("dvbs2 bb_to_soft: frame_stride shorter than header + slots");
        if (!lut_bits || lut_resolution < 2 || lut_resolution > 4096)
            throw HipError("dvbs2 bb_to_soft: the demapper table (constellation_t::make_lut) must be handed over");
        if (nframes <= 0)
            return 0;
        SD_HIP(hipSetDevice(device));
        // the stage's tables, one set per device, shared by every caller (the call is serialised: the scratch rows are part of the set); never
        // destroyed at exit -- their device buffers would be freed behind the HIP runtime's own teardown
        static std::mutex mu;
        static std::vector<S2DemapCache *> caches;
        std::lock_guard<std::mutex> lk(mu);
        if (device < 0 || device > 1023)
            throw HipError("dvbs2 bb_to_soft: device ordinal out of range");
        if ((int)caches.size() <= device)
            caches.resize(device + 1, nullptr);
        if (!caches[device])
        {
            caches[device] = new S2DemapCache;
            caches[device]->device = device;
        }
        S2DemapCache &T = *caches[device];
        if (!T.d_cw.p)
        {
            unsigned long long cw[128];
            s2_pls_codewords(cw);
            T.d_cw.reserve(128);
            SD_HIP(hipMemcpy(T.d_cw.p, cw, sizeof(cw), hipMemcpyHostToDevice));
        }
        if (T.rn_count < nsym)
        {
            std::vector<unsigned char> rn;
            s2_gold_rn(rn, 33000); // a normal QPSK frame has 32 400 data symbols (+ pilots): the longest there is
            T.d_rn.reserve(rn.size());
            SD_HIP(hipMemcpy(T.d_rn.p, rn.data(), rn.size(), hipMemcpyHostToDevice));
            T.rn_count = (int)rn.size();
        }
        const size_t lut_bytes = (size_t)lut_resolution * lut_resolution * c.bits;
        if (T.lut_host.size() != lut_bytes || memcmp(T.lut_host.data(), lut_bits, lut_bytes) != 0)
        {
            T.lut_host.assign(reinterpret_cast<const signed char *>(lut_bits), reinterpret_cast<const signed char *>(lut_bits) + lut_bytes);
            T.d_lut.reserve(lut_bytes);
            SD_HIP(hipMemcpy(T.d_lut.p, lut_bits, lut_bytes, hipMemcpyHostToDevice));
        }
        const size_t frame_soft = (size_t)nsym * c.bits;
        T.d_slots.reserve((size_t)nframes * frame_soft);
        const float2 *fr = reinterpret_cast<const float2 *>(d_plframes);
        if (d_pls)
        {
            ProfScope _ps("k_s2_pls", nullptr);
            hipLaunchKernelGGL(k_s2_pls, dim3((unsigned)nframes), dim3(64), 0, nullptr, fr, frame_stride, nframes, T.d_cw.p, d_pls);
        }
        {
            ProfScope _ps("k_s2_demap", nullptr);
            const dim3 grid((unsigned)((nsym + 255) / 256), (unsigned)nframes);
            if (c.bits == 2)
                hipLaunchKernelGGL(k_s2_demap<2>, grid, dim3(256), 0, nullptr, fr, frame_stride, nframes, nsym, pilots ? 1 : 0, T.d_rn.p, T.d_lut.p, lut_resolution, T.d_slots.p);
            else if (c.bits == 3)
                hipLaunchKernelGGL(k_s2_demap<3>, grid, dim3(256), 0, nullptr, fr, frame_stride, nframes, nsym, pilots ? 1 : 0, T.d_rn.p, T.d_lut.p, lut_resolution, T.d_slots.p);
            else
                hipLaunchKernelGGL(k_s2_demap<4>, grid, dim3(256), 0, nullptr, fr, frame_stride, nframes, nsym, pilots ? 1 : 0, T.d_rn.p, T.d_lut.p, lut_resolution, T.d_slots.p);
        }
        // the de-interleaver (S2Deinterleaver::deinterleave, dvbs2_bb_to_soft.cpp:66): frame_slot_count * 90 * bits = 64800 / 16200 soft bits per frame
        if (sdhip_s2_deinterleave_dev(device, c.constellation, shortframes ? 1 : 0, c.rate, reinterpret_cast<const int8_t *>(T.d_slots.p), d_soft, nframes) != 0)
            return -1;
        return (int)frame_soft;
        SD_GUARD_END(-1)
    }
    int64_t sdhip_s2_pl_sync_dev(int device, int slot_number, int pilots, float thresold, const float *d_syms, size_t nsyms, float *d_frames, int frame_stride,
                                 size_t max_frames, size_t *consumed, int *best_pos_out)
    {
        SD_GUARD_BEGIN
        size_t spec = 64;
        return s2_pl_sync_run(device, slot_number, pilots, thresold, d_syms, nsyms, d_frames, frame_stride, max_frames, consumed, best_pos_out, &spec);
        SD_GUARD_END(-1)
    }
}
namespace sdhip
{
    // spec_io: how many frame windows the next launch speculates on, carried by a caller with a stream (the engine): a stream in lock searches all
    // the windows of a call in ONE launch
    int64_t s2_pl_sync_run(int device, int slot_number, int pilots, float thresold, const float *d_syms, size_t nsyms, float *d_frames, int frame_stride, size_t max_frames,
                           size_t *consumed, int *best_pos_out, size_t *spec_io)
    {
        if (slot_number <= 0 || slot_number > 360)
            throw HipError("dvbs2 pl_sync: slot_number out of range");
        const int raw = s2_raw_frame_size(slot_number, pilots);
        if (frame_stride < raw)
            throw HipError("dvbs2 pl_sync: frame_stride shorter than the raw frame");
        if (consumed)
            *consumed = 0;
        SD_HIP(hipSetDevice(device));
        const float2 *sy = reinterpret_cast<const float2 *>(d_syms);
        // Speculate-and-certify over the frame chain: frame k's window starts where frame k-1's ended (raw + its best_pos further). In lock every
        // best_pos is 0, so the windows of a whole batch are known in advance: search them all in one launch, accept the run of zeros and the first
        // frame behind it, continue from there. After a miss the next launch speculates on fewer frames (a stream of noise costs a launch per
        // frame either way).
        std::vector<long long> starts;
        std::vector<int> bps;
        DevBuf<int> d_bp;
        std::vector<int> h_bp;
        long long pos = 0; // read position of the block's ring buffer
        size_t spec = std::max<size_t>(8, *spec_io);
        while (starts.size() < max_frames)
        {
            const long long avail = (long long)nsyms - pos;
            long long can = avail / raw; // frames whose WINDOW is there
            if (can <= 0)
                break;
            const size_t nspec = (size_t)std::min<long long>(std::min<long long>(can, (long long)spec), (long long)(max_frames - starts.size()));
            d_bp.reserve(nspec);
            h_bp.resize(nspec);
            {
                ProfScope _ps("k_s2_plsync_search", nullptr);
                hipLaunchKernelGGL(k_s2_plsync_search, dim3((unsigned)nspec), dim3(256), 0, nullptr, sy, pos, raw, (int)nspec, thresold, d_bp.p);
            }
            SD_HIP(hipMemcpy(h_bp.data(), d_bp.p, nspec * sizeof(int), hipMemcpyDeviceToHost));
            size_t k = 0;
            bool stop = false;
            for (; k < nspec; k++)
            {
                const int bp = h_bp[k];
                // work2 reads the frame, then best_pos more symbols to re-align (:112-118): both must be there
                if (pos + raw + bp > (long long)nsyms)
                {
                    stop = true;
                    break;
                }
                starts.push_back(pos + bp);
                bps.push_back(bp);
                pos += raw + bp;
                if (bp != 0)
                {
                    k++;
                    break; // the windows behind this frame were speculated at the wrong place
                }
            }
            if (stop)
                break;
            spec = (k == nspec) ? std::min<size_t>(spec * 2, 65536) : 8;
        }
        *spec_io = spec;
        const size_t nf = starts.size();
        if (nf > 0)
        {
            DevBuf<long long> d_st;
            d_st.reserve(nf);
            SD_HIP(hipMemcpy(d_st.p, starts.data(), nf * sizeof(long long), hipMemcpyHostToDevice));
            ProfScope _ps("k_s2_plsync_emit", nullptr);
            hipLaunchKernelGGL(k_s2_plsync_emit, dim3((unsigned)((raw + 255) / 256), (unsigned)nf), dim3(256), 0, nullptr, sy, d_st.p, raw, (int)nf,
                               reinterpret_cast<float2 *>(d_frames), frame_stride);
            SD_HIP(hipDeviceSynchronize());
        }
        if (consumed)
            *consumed = (size_t)pos;
        if (best_pos_out)
            for (size_t k = 0; k < nf; k++)
                best_pos_out[k] = bps[k];
        return (int64_t)nf;
    }
} // namespace sdhip
extern "C"
{
    int sdhip_s2_pll_dev(int device, int modcod, int shortframes, int pilots, float loop_bw, const float *d_frames_in, float *d_frames_out, int frame_stride, int nframes,
                         const float *lut_phase_error, int lut_resolution, float *state2)
    {
        return sdhip_s2_pll_frames_dev(device, modcod, shortframes, pilots, loop_bw, d_frames_in, d_frames_out, frame_stride, nframes, lut_phase_error, lut_resolution, state2, 1, nullptr);
    }
    int sdhip_s2_pll_frames_dev(int device, int modcod, int shortframes, int pilots, float loop_bw, const float *d_frames_in, float *d_frames_out, int frame_stride, int nframes,
                                const float *lut_phase_error, int lut_resolution, float *state2, int mode, unsigned *stats4)
    {
        SD_GUARD_BEGIN
        if (!state2)
            throw HipError("dvbs2 pll: the demapper table's phase errors and the loop state must be handed over");
        // the table is the caller's data: keep the last runner per calling thread while configuration and table stay the same (a stream of calls
        // with carried state then uploads nothing); an engine handle (sdhip_dvbs2_demod_create) owns its own
        struct Key
        {
            int device, modcod, sf, pilots, res, exact;
            float bw;
            std::vector<float> lut;
        };
        static std::mutex mu;
        static S2Pll *runner = nullptr; // never destroyed at exit: its device buffers would be freed behind the HIP runtime's own teardown
        static Key key;
        std::lock_guard<std::mutex> lk(mu);
        const size_t nl = (lut_phase_error && lut_resolution >= 2 && lut_resolution <= 4096) ? (size_t)lut_resolution * lut_resolution : 0;
        const bool same = runner && key.device == device && key.modcod == modcod && key.sf == (shortframes ? 1 : 0) && key.pilots == (pilots ? 1 : 0) && key.res == lut_resolution &&
                          key.bw == loop_bw && key.lut.size() == nl && nl && memcmp(key.lut.data(), lut_phase_error, nl * sizeof(float)) == 0;
        if (!same)
        {
            delete runner;
            runner = nullptr;
            runner = new S2Pll(device, modcod, shortframes, pilots, loop_bw, lut_phase_error, lut_resolution, true);
            key = Key{device, modcod, shortframes ? 1 : 0, pilots ? 1 : 0, lut_resolution, 1, loop_bw, std::vector<float>(lut_phase_error, lut_phase_error + nl)};
        }
        runner->exact = mode == 1;
        runner->state = S2PllState{state2[0], state2[1]};
        // mode 0: the frame-parallel schedule with the caller vouching that state2 is a locked loop's (any call but a stream's first); mode 2: the
        // same on a new stream (the first frames are walked serially)
        runner->have_hint = mode == 0;
        if (mode == 0)
            runner->set_hint(state2[1]);
        if (nframes > 0)
            runner->run(d_frames_in, d_frames_out, frame_stride, nframes, nullptr);
        state2[0] = runner->state.phase;
        state2[1] = runner->state.freq;
        if (stats4)
        {
            stats4[0] = runner->stats.lanes;
            stats4[1] = runner->stats.rerun;
            stats4[2] = runner->stats.forced;
            stats4[3] = runner->stats.serial_frames;
        }
        return runner->per_frame();
        SD_GUARD_END(-1)
    }
    int sdhip_op_atan2f(int device, const float *d_y, const float *d_x, int n, float *d_out)
    {
        SD_GUARD_BEGIN
        SD_HIP(hipSetDevice(device));
        hipLaunchKernelGGL(k_s2_atan2f, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_y, d_x, n, d_out);
        SD_HIP(hipDeviceSynchronize());
        return 0;
        SD_GUARD_END(-1)
    }
    int sdhip_s2_cfg(int modcod, int shortframes, int *bits, int *slots, int *rate, int *constellation)
    {
        SD_GUARD_BEGIN
        const S2Cfg c = s2_cfg_of(modcod, shortframes ? 1 : 0);
        *bits = c.bits;
        *slots = c.slots;
        *rate = c.rate;
        *constellation = c.constellation;
        return 0;
        SD_GUARD_END(-1)
    }
}
