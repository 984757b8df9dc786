// study harness: M&M loop (same arithmetic as clock_recovery_mm.cpp) with a gear-shifted warm-up; logs symbol times
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float re, im; } cf_t;
static float clipf(float x, float c) { return x < -c ? -c : (x > c ? c : x); }
// in: samples (with 8 samples of valid history before index 0 NOT required: start>=8)
// returns number of symbols; t_out[m] = absolute time (inc+mu) of symbol m BEFORE the update (the interpolation instant), om_out = omega after
long mm_track(const cf_t *in, long n, long start, const float *bank, float omega_mid, float omega_gain, float mu_gain, float rel_limit,
              long n_fast, float G, float Gom, long n_total, double *t_out, float *om_out, int freeze_omega_fast)
{
    float mu = 0.5f, omega = omega_mid, lim = rel_limit * omega_mid;
    cf_t p2 = {0, 0}, p1 = {0, 0}, p0 = {0, 0}, c2 = {0, 0}, c1 = {0, 0}, c0 = {0, 0};
    long inc = start, m = 0;
    while (m < n_total && inc + 8 < n)
    {
        float mg = m < n_fast ? mu_gain * G : mu_gain;
        float og = m < n_fast ? (freeze_omega_fast ? 0.0f : omega_gain * Gom) : omega_gain;
        p2 = p1; p1 = p0; c2 = c1; c1 = c0;
        int imu = (int)rint(mu * 128);
        if (imu < 0) imu = 0;
        if (imu >= 128) imu = 127;
        const float *t = bank + imu * 8;
        float re = 0, im = 0;
        for (int k = 0; k < 8; k++) { re += in[inc - 7 + k].re * t[k]; im += in[inc - 7 + k].im * t[k]; }
        p0.re = re; p0.im = im;
        c0.re = re > 0 ? 1.f : 0.f; c0.im = im > 0 ? 1.f : 0.f;
        float ur = p0.re - p2.re, ui = p0.im - p2.im;
        float a_re = ur * c1.re - ui * (-c1.im);
        float vr = c0.re - c2.re, vi = c0.im - c2.im;
        float b_re = vr * p1.re - vi * (-p1.im);
        float pe = clipf(a_re - b_re, 1.0f);
        t_out[m] = (double)inc + mu;
        omega = omega + og * pe;
        omega = omega_mid + clipf(omega - omega_mid, lim);
        om_out[m] = omega;
        mu = mu + omega + mg * pe;
        float fl = floorf(mu);
        inc += (long)fl;
        mu -= fl;
        m++;
    }
    return m;
}
