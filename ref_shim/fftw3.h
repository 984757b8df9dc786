/* ref_shim/fftw3.h -- declarations only. The reference's dsp/filter/rrc.h includes dsp/filter/fft.h (the FFT-filter variant of the
 * RRC block), which names these types in its members; nothing we compile instantiates it, so no definition is ever needed. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
typedef float fftwf_complex[2];
typedef struct sdshim_fftwf_plan_s *fftwf_plan;
#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_ESTIMATE (1U << 6)
void *fftwf_malloc(unsigned long n);
void fftwf_free(void *p);
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);
#ifdef __cplusplus
}
#endif
