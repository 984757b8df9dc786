// dsp_design.h -- host-side filter design and loop-constant derivation for psk_demod (double precision,
// evaluated once per handle). Formulas follow the reference's designers so that the coefficient tables the
// kernels consume are the same numbers the reference's blocks are constructed with:
//   root_raised_cosine   src-core/common/dsp/filter/firdes.cpp:34-78
//   windowed_sinc/nuttall src-core/common/dsp/window/window.cpp:9-50   (M&M 128x8 interpolator bank)
//   low_pass + Kaiser     src-core/common/dsp/filter/firdes.cpp:80-120, 355-480 (rational resampler prototype)
//   polyphase layout      src-core/common/dsp/resamp/polyphase_bank.cpp:6-39
//   Costas alpha/beta     src-core/common/dsp/pll/costas_loop.cpp:5-12
#pragma once
#include <cmath>
#include <numeric>
#include <vector>

namespace sdhip
{
    namespace design
    {
        constexpr double PI = 3.14159265358979323846;

        inline std::vector<float> rrc(double gain, double fs, double symrate, double alpha, int ntaps)
        {
            ntaps |= 1;
            const double spb = fs / symrate;
            std::vector<float> taps(ntaps);
            double scale = 0;
            for (int i = 0; i < ntaps; i++)
            {
                double x1, x2, x3, num, den;
                const double xindx = i - ntaps / 2;
                x1 = PI * xindx / spb;
                x2 = 4 * alpha * xindx / spb;
                x3 = x2 * x2 - 1;
                if (std::fabs(x3) >= 0.000001)
                {
                    if (i != ntaps / 2)
                        num = std::cos((1 + alpha) * x1) + std::sin((1 - alpha) * x1) / (4 * alpha * xindx / spb);
                    else
                        num = std::cos((1 + alpha) * x1) + (1 - alpha) * PI / (4 * alpha);
                    den = x3 * PI;
                }
                else
                {
                    if (alpha == 1)
                    {
                        taps[i] = -1;
                        scale += taps[i];
                        continue;
                    }
                    x3 = (1 - alpha) * x1;
                    x2 = (1 + alpha) * x1;
                    num = (std::sin(x2) * (1 + alpha) * PI - std::cos(x3) * ((1 - alpha) * PI * spb) / (4 * alpha * xindx) +
                           std::sin(x3) * spb * spb / (4 * alpha * xindx * xindx));
                    den = -32 * PI * alpha * alpha * xindx / spb;
                }
                taps[i] = (float)(4 * alpha * num / den);
                scale += taps[i];
            }
            for (int i = 0; i < ntaps; i++)
                taps[i] = (float)(taps[i] * gain / scale);
            return taps;
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

        inline double izero(double x)
        {
            double sum, u, halfx, temp;
            int n;
            sum = u = n = 1;
            halfx = x / 2.0;
            do
            {
                temp = halfx / (double)n;
                n += 1;
                temp *= temp;
                u *= temp;
                sum += u;
            } while (u >= 1E-21 * sum);
            return sum;
        }

        // design_resampler_filter_float + low_pass (Kaiser beta 7, fractional_bw 0.4); interp/decim reduced by their gcd
        inline int resampler_bank(unsigned &interpolation, unsigned &decimation, std::vector<float> &bank)
        {
            const unsigned g = std::gcd(interpolation, decimation);
            interpolation /= g;
            decimation /= g;
            const float fractional_bw = 0.4f;
            float beta = 7.0;
            float halfband = 0.5;
            float rate = float(interpolation) / float(decimation);
            float trans_width, mid_transition_band;
            if (rate >= 1.0)
            {
                trans_width = halfband - fractional_bw;
                mid_transition_band = (float)(halfband - trans_width / 2.0);
            }
            else
            {
                trans_width = rate * (halfband - fractional_bw);
                mid_transition_band = (float)(rate * halfband - trans_width / 2.0);
            }
            double gain = interpolation, sampling_freq = interpolation, cutoff_freq = mid_transition_band, transition_width = trans_width;
            const double att = (double)beta / 0.1102 + 8.7;
            int ntaps = (int)(att * sampling_freq / (22.0 * transition_width));
            if ((ntaps & 1) == 0)
                ntaps++;
            std::vector<float> taps(ntaps), w(ntaps);
            {
                const double IBeta = 1.0 / izero(beta);
                const double inm1 = 1.0 / ((double)(ntaps - 1));
                w[0] = (float)IBeta;
                for (int i = 1; i < ntaps - 1; i++)
                {
                    const double temp = 2 * i * inm1 - 1;
                    w[i] = (float)(izero(beta * std::sqrt(1.0 - temp * temp)) * IBeta);
                }
                w[ntaps - 1] = (float)IBeta;
            }
            const int M = (ntaps - 1) / 2;
            const double fwT0 = 2 * PI * cutoff_freq / sampling_freq;
            for (int n = -M; n <= M; n++)
            {
                if (n == 0)
                    taps[n + M] = (float)(fwT0 / PI * w[n + M]);
                else
                    taps[n + M] = (float)(std::sin(n * fwT0) / (n * PI) * w[n + M]);
            }
            double fmax = taps[0 + M];
            for (int n = 1; n <= M; n++)
                fmax += 2 * taps[n + M];
            gain /= fmax;
            for (int i = 0; i < ntaps; i++)
                taps[i] = (float)(taps[i] * gain);
            return polyphase(taps, (int)interpolation, bank);
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
