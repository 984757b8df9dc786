// dsp_design.h -- host-side filter design and loop-constant derivation for psk_demod (double precision,
// evaluated once per handle). Formulas follow the reference's designers so that the coefficient tables the
// kernels consume are the same numbers the reference's blocks are constructed with:
//   root_raised_cosine   src-core/common/dsp/filter/firdes.cpp:34-78
//   windowed_sinc/nuttall src-core/common/dsp/window/window.cpp:9-50   (M&M 128x8 interpolator bank)
//   low_pass + Kaiser     src-core/common/dsp/filter/firdes.cpp:80-120, 355-480 (rational resampler prototype)
//   polyphase layout      src-core/common/dsp/resamp/polyphase_bank.cpp:6-39
//   Costas alpha/beta     src-core/common/dsp/pll/costas_loop.cpp:5-12
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace sdhip
{
    namespace design
    {
        constexpr double PI = 3.14159265358979323846;

        // ---- root-raised-cosine matched filter -----------------------------------------------------------------
        // The pulse h(t) = [cos((1+a) pi t) + sin((1-a) pi t) / (4 a t)] * 4a / (pi ((4 a t)^2 - 1)), t in symbols, sampled at
        // t = k / sps for k = -(n-1)/2 .. (n-1)/2, each tap rounded to float, then all of them scaled to unit sum. The reference
        // evaluates it in double in one particular operand order (firdes.cpp:34-78); the taps here must be those floats bit for
        // bit, so each sub-expression below keeps the reference's association -- e.g. (pi * k) / sps, never pi * (k / sps) --
        // while the organisation is ours: one function per branch of the formula, the pulse's even symmetry used to evaluate
        // only one half (sin and cos are odd / even to the last bit, so the mirrored tap is the same float), the centre tap on
        // its own.
        struct RrcPulse
        {
            double sps, a; // samples per symbol, roll-off
            // regular point k != 0 (the denominator (4 a k / sps)^2 - 1 is away from zero)
            double regular(double k, double pole_term) const
            {
                const double ph = PI * k / sps;
                const double top = std::cos((1 + a) * ph) + std::sin((1 - a) * ph) / (4 * a * k / sps);
                return 4 * a * top / (pole_term * PI);
            }
            // centre: sin(x)/x -> its limit
            double centre(double pole_term) const
            {
                const double ph = PI * 0.0 / sps;
                const double top = std::cos((1 + a) * ph) + (1 - a) * PI / (4 * a);
                return 4 * a * top / (pole_term * PI);
            }
            // on the pole 4 a k / sps = +-1 the closed form is 0/0: l'Hopital's form, as the reference writes it
            double on_pole(double k) const
            {
                const double ph = PI * k / sps;
                const double lo = (1 - a) * ph, hi = (1 + a) * ph;
                const double top = std::sin(hi) * (1 + a) * PI - std::cos(lo) * ((1 - a) * PI * sps) / (4 * a * k) + std::sin(lo) * sps * sps / (4 * a * k * k);
                return 4 * a * top / (-32 * PI * a * a * k / sps);
            }
            float tap(int k) const
            {
                const double kk = (double)k;
                const double q = 4 * a * kk / sps;
                const double pole_term = q * q - 1;
                if (std::fabs(pole_term) < 0.000001)
                    return a == 1 ? -1.0f : (float)on_pole(kk);
                return (float)(k == 0 ? centre(pole_term) : regular(kk, pole_term));
            }
        };
        inline std::vector<float> rrc(double gain, double fs, double symrate, double alpha, int ntaps)
        {
            const int n = ntaps | 1, mid = n / 2;
            const RrcPulse pulse{fs / symrate, alpha};
            std::vector<float> h(n);
            h[mid] = pulse.tap(0);
            for (int k = 1; k <= mid; k++)
                h[mid - k] = h[mid + k] = pulse.tap(-k); // evaluated on the negative side, where the reference's loop meets the value first
            double sum = 0; // the reference adds the float taps up in index order, in double
            for (float v : h)
                sum += v;
            for (float &v : h)
                v = (float)(v * gain / sum);
            return h;
        }

        // bank[(nfilt-1) - (i % nfilt)][i / nfilt] = proto[i]; returns taps per phase
        inline int polyphase(const std::vector<float> &proto, int nfilt, std::vector<float> &bank)
        {
            const int n = (int)proto.size();
            int ntaps = (n + nfilt - 1) / nfilt;
            if (std::fmod((double)n / (double)nfilt, 1.0) > 0.0)
                ntaps++;
            bank.assign((size_t)nfilt * ntaps, 0.0f);
            for (int i = 0; i < nfilt * ntaps; i++)
                bank[(size_t)((nfilt - 1) - (i % nfilt)) * ntaps + i / nfilt] = (i < n) ? proto[i] : 0.0f;
            return ntaps;
        }

        inline int mm_bank(int nfilt, int ntaps, std::vector<float> &bank)
        {
            const int count = nfilt * ntaps;
            std::vector<float> proto(count);
            const double omega = 2.0 * PI * ((0.5 / (double)nfilt) / 1.0);
            const double half = (double)count / 2.0;
            const double corr = (double)nfilt * omega / PI;
            static const double coefs[] = {0.355768, 0.487396, 0.144232, 0.012604};
            for (int i = 0; i < count; i++)
            {
                const double t = (double)i - half + 0.5;
                const double xs = t * omega;
                const double sinc = (xs == 0.0) ? 1.0 : (std::sin(xs) / xs);
                double win = 0.0, sign = 1.0;
                for (int c = 0; c < 4; c++)
                {
                    win += sign * coefs[c] * std::cos((double)c * 2.0 * PI * (t - half) / (double)count);
                    sign = -sign;
                }
                proto[i] = (float)(sinc * win * corr);
            }
            return polyphase(proto, nfilt, bank);
        }

        // ---- rational resampler prototype -----------------------------------------------------------------------
        // Kaiser-windowed sinc low-pass at the narrower of the two Nyquist bands (design_resampler_filter_float -> low_pass(...,
        // WIN_KAISER, beta = 7), firdes.cpp:276-301, 80-120; Kaiser window fft/window.cpp). float / double conversions sit where the
        // reference's declarations put them: they decide the tap COUNT (a truncation) and the last bits of every tap.

        // I0(x) by its power series sum_k ((x/2)^k / k!)^2, terms added until they vanish against the sum
        inline double bessel_i0(double x)
        {
            const double h = x / 2.0;
            double term = 1.0, total = 1.0;
            for (int k = 1;; k++)
            {
                double q = h / (double)k;
                q *= q;
                term *= q;
                total += term;
                if (!(term >= 1E-21 * total))
                    return total;
            }
        }
        // w[i] = I0(beta sqrt(1 - (2i/(n-1) - 1)^2)) / I0(beta); the end points are 1 / I0(beta) by definition
        inline std::vector<float> kaiser_window(int n, double beta)
        {
            const double norm = 1.0 / bessel_i0(beta), step = 1.0 / ((double)(n - 1));
            std::vector<float> w(n, (float)norm);
            for (int i = 1; i + 1 < n; i++)
            {
                const double u = 2 * i * step - 1;
                w[i] = (float)(bessel_i0(beta * std::sqrt(1.0 - u * u)) * norm);
            }
            return w;
        }
        // length from the window's attenuation (beta / 0.1102 + 8.7 dB) and the transition width, forced odd; ideal low-pass
        // sin(n wc) / (n pi) under the window; DC gain normalised to `gain`
        inline std::vector<float> kaiser_lowpass(double gain, double fs, double cutoff, double width, double beta)
        {
            int n = (int)((beta / 0.1102 + 8.7) * fs / (22.0 * width));
            n |= 1;
            const std::vector<float> w = kaiser_window(n, beta);
            const int half = (n - 1) / 2;
            const double wc = 2 * PI * cutoff / fs;
            std::vector<float> h(n);
            h[half] = (float)(wc / PI * w[half]);
            for (int m = 1; m <= half; m++)
            {
                h[half + m] = (float)(std::sin(m * wc) / (m * PI) * w[half + m]);
                h[half - m] = (float)(std::sin(-m * wc) / (-m * PI) * w[half - m]);
            }
            double dc = h[half];
            for (int m = 1; m <= half; m++)
                dc += 2 * h[half + m];
            const double k = gain / dc;
            for (float &v : h)
                v = (float)(v * k);
            return h;
        }
        // interp / decim reduced by their gcd; pass band 0.4 of the narrower Nyquist band, transition up to that band's edge;
        // the bank is the polyphase split of the prototype over `interp` arms
        inline int resampler_bank(unsigned &interpolation, unsigned &decimation, std::vector<float> &bank)
        {
            const unsigned g = std::gcd(interpolation, decimation);
            interpolation /= g;
            decimation /= g;
            const float passband = 0.4f, nyquist = 0.5f, beta = 7.0f;
            const float ratio = float(interpolation) / float(decimation);
            const float shrink = ratio >= 1.0 ? 1.0f : ratio; // decimating: everything scales with the output band
            float width, centre;
            if (ratio >= 1.0)
            {
                width = nyquist - passband;
                centre = (float)(nyquist - width / 2.0);
            }
            else
            {
                width = shrink * (nyquist - passband);
                centre = (float)(shrink * nyquist - width / 2.0);
            }
            const std::vector<float> proto = kaiser_lowpass((double)interpolation, (double)interpolation, centre, width, beta);
            return polyphase(proto, (int)interpolation, bank);
        }

        // the arctangent table of the carrier PLL's phase detector (fast_trig.cpp:16-59): atan(i / 255) as the seven-significant-digit
        // decimal literals of the reference's source, read as float; entry 256 repeats entry 255 (the interpolation's upper neighbour
        // at ratio 1)
        inline std::vector<float> atan_table()
        {
            std::vector<float> t(257);
            for (int i = 0; i < 257; i++)
            {
                char txt[32];
                snprintf(txt, sizeof(txt), "%.6e", std::atan((double)std::min(i, 255) / 255.0));
                t[i] = strtof(txt, nullptr);
            }
            return t;
        }

        inline void costas_gains(float loop_bw, float &alpha, float &beta)
        {
            const float damping = sqrtf(2.0f) / 2.0f;
            const float denom = (float)(1.0 + 2.0 * damping * loop_bw + loop_bw * loop_bw);
            alpha = (4 * damping * loop_bw) / denom;
            beta = (4 * loop_bw * loop_bw) / denom;
        }
    } // namespace design
} // namespace sdhip
