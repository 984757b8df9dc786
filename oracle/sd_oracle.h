/* oracle/sd_oracle.h -- TEST INFRASTRUCTURE ONLY (see sd_oracle.c). */
#ifndef SD_ORACLE_H
#define SD_ORACLE_H
#include <stdint.h>
#include "../include/sdhip.h"
#ifdef __cplusplus
extern "C" {
#endif

void sdo_ccdecoder(int frame_bits, const uint8_t *syms, int nblocks, uint8_t *out);
void sdo_ccencode(const uint8_t *bits, int nbits, uint8_t *out);
void sdo_derand(uint8_t *data, int len);
void sdo_rs_decode(uint8_t *data, int nframes, int frame_stride, int dualbasis, int I, int rs239, int fill_bytes, int *errors);
int sdo_deframer(const uint8_t *bits, int64_t nbits, int chunk, int cadu_size, uint32_t asm_sync, int state_synced, uint8_t *out, int64_t out_cap_frames);
int64_t sdo_concat_decode(const sdhip_fec_cfg *c, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                          uint8_t *vit_bits, int64_t *vit_nbits, float *blk_ber, int *blk_state, int *frm_err, int64_t *n_deframed);
/* conv_rate of ccsds_conv_concat_decoder other than "1/2" (Viterbi_Depunc, SURVEY.md 8 row a13'); rate = SDO_RATE_* */
enum { SDO_RATE_2_3 = 1, SDO_RATE_3_4 = 2, SDO_RATE_5_6 = 3, SDO_RATE_7_8 = 4 };
int64_t sdo_concat_decode_punc(const sdhip_fec_cfg *c, int rate, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                               uint8_t *vit_bits, int64_t *vit_nbits, float *blk_ber, int *blk_state, int *frm_err, int64_t *n_deframed);
int64_t sdo_simple_decode(const sdhip_fec_cfg *c, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames, int *frm_err, int64_t *n_deframed);
int64_t sdo_metop_decode(float ber_thr, int outsync_after, const int8_t *soft, int64_t n, uint8_t *cadu_out, int64_t cadu_cap_frames,
                         uint8_t *vit_bits, int64_t *vit_nbits, float *blk_ber, int *blk_state, int *frm_err);

int sdo_rrc_taps(double gain, double fs, double symrate, double alpha, int ntaps, float *out);
int sdo_mm_bank(int nfilt, int ntaps, float *out);
int sdo_resamp_bank(unsigned interp, unsigned decim, float *out, int cap, int *interp_red, int *decim_red);
int64_t sdo_block_run(int kind, const float *p, const float *in_c, int64_t n, int chunk, float *out_c, int64_t out_cap);
/* test tap of the M&M restatement: see sd_oracle_dsp.inc */
void sdo_mm_set_tap(int64_t *buf, int64_t cap);
int64_t sdo_mm_tap_count(void);
int64_t sdo_psk_demod(const sdhip_demod_cfg *c, const float *iq, int64_t n, int8_t *soft, int64_t soft_cap, float *syms, int64_t syms_cap,
                      int *buffer_size_out, float *final_sps_out);
/* dsp::DopplerCorrectBlock::work's sample loop (src-core/common/dsp/utils/doppler_correct.cpp:41-63), restated; the target frequency the block
   recomputes behind every source buffer (:68-93, SGP4 on a TLE from SatDump's database) is an INPUT here: targets[k] is in force during source buffer
   k + 1, the first buffer runs on 0. state2 = {phase, freq} in / out. Pinned against the block itself, compiled in place with libpredict
   (oracle/ref_wrap_doppler.cpp, tests/test_oracle_vs_ref.py::test_doppler_restatement_equals_the_block). */
void sdo_doppler(const float *in_c, int64_t n, float alpha, int buf_len, const float *targets, int ntargets, float *out_c, float *state2);
/* glibc-2.35 sinf/cosf restatement (the device code must match this bit for bit) */
float sdo_sinf(float x);
float sdo_cosf(float x);
#ifdef __cplusplus
}
#endif
#endif
