// Minimal stand-in for <volk/volk.h> so that the reference's hot-path sources
// (under /root/reference/src-core) compile unmodified in this container, which
// has no libvolk. TEST INFRASTRUCTURE ONLY (oracle/_ref build) - never linked
// into the product library.
//
// Every kernel here is the strict sequential ("_generic") accumulation order.
// <math.h> is included on purpose: real VOLK's volk_common.h pulls it in, which
// makes the unqualified sqrt(float) in common/dsp/utils/agc.cpp:32 resolve to
// the float overload ("AGC sqrt overload trap").
#pragma once
#include <cstdlib>
#include <cstddef>
#include <cstdint>
#include <complex>
#include <cstring>
#include <math.h>
#define VOLK_VERSION 030200
typedef std::complex<float> lv_32fc_t;
static inline size_t volk_get_alignment() { return 32; }
static inline void *volk_malloc(size_t size, size_t align)
{
    void *p = nullptr;
    if (posix_memalign(&p, align < sizeof(void *) ? sizeof(void *) : align, size ? size : align))
        return nullptr;
    return p;
}
static inline void volk_free(void *p) { free(p); }
static inline void volk_32fc_32f_dot_prod_32fc(lv_32fc_t *r, const lv_32fc_t *in, const float *t, unsigned n)
{
    float re = 0, im = 0;
    const float *a = (const float *)in;
    for (unsigned i = 0; i < n; i++)
    {
        re += a[2 * i] * t[i];
        im += a[2 * i + 1] * t[i];
    }
    *r = lv_32fc_t(re, im);
}
#define volk_32fc_32f_dot_prod_32fc_a volk_32fc_32f_dot_prod_32fc
static inline void volk_32f_x2_dot_prod_32f(float *r, const float *in, const float *t, unsigned n)
{
    float s = 0;
    for (unsigned i = 0; i < n; i++)
        s += in[i] * t[i];
    *r = s;
}
#define volk_32f_x2_dot_prod_32f_a volk_32f_x2_dot_prod_32f
// VOLK's _generic kernel: sqrtf((real * real) + (imag * imag)) per point (dsp/agc/agc_fast.cpp:37)
static inline void volk_32fc_magnitude_32f(float *m, const lv_32fc_t *in, unsigned n)
{
    const float *a = (const float *)in;
    for (unsigned i = 0; i < n; i++)
        m[i] = sqrtf((a[2 * i] * a[2 * i]) + (a[2 * i + 1] * a[2 * i + 1]));
}
/* complex taps (the ndsp FIR block's third instantiation, dsp/filter/fir.cpp:122): VOLK's generic kernel, one complex MAC per point */
static inline void volk_32fc_x2_dot_prod_32fc(lv_32fc_t *r, const lv_32fc_t *in, const lv_32fc_t *t, unsigned n)
{
    const float *a = (const float *)in, *b = (const float *)t;
    float re = 0, im = 0;
    for (unsigned i = 0; i < n; i++)
    {
        re += a[2 * i] * b[2 * i] - a[2 * i + 1] * b[2 * i + 1];
        im += a[2 * i] * b[2 * i + 1] + a[2 * i + 1] * b[2 * i];
    }
    *r = lv_32fc_t(re, im);
}
#define volk_32fc_x2_dot_prod_32fc_a volk_32fc_x2_dot_prod_32fc
/* element-wise complex kernels (dvbs2_pl_sync.cpp:77-78): VOLK's generic forms */
static inline void volk_32fc_conjugate_32fc(lv_32fc_t *out, const lv_32fc_t *in, unsigned n)
{
    const float *a = (const float *)in;
    float *o = (float *)out;
    for (unsigned i = 0; i < n; i++)
    {
        o[2 * i] = a[2 * i];
        o[2 * i + 1] = -a[2 * i + 1];
    }
}
static inline void volk_32fc_x2_multiply_32fc(lv_32fc_t *out, const lv_32fc_t *x, const lv_32fc_t *y, unsigned n)
{
    const float *a = (const float *)x, *b = (const float *)y;
    float *o = (float *)out;
    for (unsigned i = 0; i < n; i++)
    {
        const float re = a[2 * i] * b[2 * i] - a[2 * i + 1] * b[2 * i + 1];
        const float im = a[2 * i] * b[2 * i + 1] + a[2 * i + 1] * b[2 * i];
        o[2 * i] = re;
        o[2 * i + 1] = im;
    }
}
static inline void volk_16i_s32f_convert_32f_u(float *o, const int16_t *in, float s, unsigned n)
{
    const float is = 1.0f / s;
    for (unsigned i = 0; i < n; i++)
        o[i] = (float)in[i] * is;
}
static inline void volk_8i_s32f_convert_32f_u(float *o, const int8_t *in, float s, unsigned n)
{
    const float is = 1.0f / s;
    for (unsigned i = 0; i < n; i++)
        o[i] = (float)in[i] * is;
}
static inline void volk_32i_s32f_convert_32f_u(float *o, const int32_t *in, float s, unsigned n)
{
    const float is = 1.0f / s;
    for (unsigned i = 0; i < n; i++)
        o[i] = (float)in[i] * is;
}
struct volk_func_desc
{
    const char **impl_names;
    const int *impl_deps;
    const bool *impl_alignment;
    size_t n_impls;
};
static inline volk_func_desc volk_8u_x4_conv_k7_r2_8u_get_func_desc()
{
    static const char *names[] = {"generic"};
    return volk_func_desc{names, nullptr, nullptr, 1};
}
static inline void volk_8u_x4_conv_k7_r2_8u_manual(unsigned char *, unsigned char *, unsigned char *, unsigned char *, unsigned, unsigned, unsigned char *, const char *) { abort(); }

// volk_32fc_s32fc_x2_rotator2_32fc, VOLK >= 3.1 (kernels/volk/volk_32fc_s32fc_x2_rotator2_32fc.h, _generic): out = in * phase, phase *= inc
// per sample, the phase renormalised every ROTATOR_RELOAD = 512 samples and once more at the end of every call that had a remainder.
// VOLK is a system library outside the reference tree; this is its published generic kernel, float complex products
// in the plain four-multiply form, hypotf from libm. (VOLK's SIMD variants order the products differently: the oracle pins the generic one.)
static inline void volk_32fc_s32fc_x2_rotator2_32fc(lv_32fc_t *out, const lv_32fc_t *in, const lv_32fc_t *phase_inc, lv_32fc_t *phase, unsigned int num_points)
{
    unsigned int i = 0;
    for (i = 0; i < num_points / 512; ++i)
    {
        for (int j = 0; j < 512; ++j)
        {
            *out++ = *in++ * (*phase);
            (*phase) *= *phase_inc;
        }
        (*phase) /= hypotf(phase->real(), phase->imag());
    }
    for (i = 0; i < num_points % 512; ++i)
    {
        *out++ = *in++ * (*phase);
        (*phase) *= *phase_inc;
    }
    if (i)
        (*phase) /= hypotf(phase->real(), phase->imag());
}
