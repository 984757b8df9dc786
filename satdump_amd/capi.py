"""ctypes binding of the C ABI in include/sdhip.h (libsdhip.so). Plumbing only: no compute happens here.

The library is REQUIRED: importing this module on a machine without the built extension raises, and every
call into it fails loudly when no HIP device is present -- there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDHIP_LIB") or os.path.join(_HERE, "lib", "libsdhip.so")  # SDHIP_LIB: experiment builds only

BPSK, BPSK_90, QPSK, OQPSK, PSK8 = 0, 1, 2, 3, 4
RS_NONE, RS223, RS239 = 0, 1, 2
RATE_1_2, RATE_2_3, RATE_3_4, RATE_5_6, RATE_7_8 = 0, 1, 2, 3, 4
FMT_CF32, FMT_CS16, FMT_CS8, FMT_CU8, FMT_CS32 = 0, 1, 2, 3, 4
DEC_CONV_CONCAT, DEC_METOP_AHRPT, DEC_SIMPLE_PSK, DEC_FENGYUN_AHRPT, DEC_FENGYUN_MPT = 0, 1, 2, 3, 4
CONSTELLATIONS = {"bpsk": BPSK, "bpsk_90": BPSK_90, "qpsk": QPSK, "oqpsk": OQPSK, "8psk": PSK8}


class DemodCfg(C.Structure):
    _fields_ = [
        ("samplerate", C.c_double), ("symbolrate", C.c_double), ("constellation", C.c_int), ("rrc_alpha", C.c_float),
        ("rrc_taps", C.c_int), ("pll_bw", C.c_float), ("agc_rate", C.c_float), ("dc_block", C.c_int), ("iq_swap", C.c_int),
        ("min_sps", C.c_float), ("max_sps", C.c_float), ("clock_gain_omega", C.c_float), ("clock_mu", C.c_float),
        ("clock_gain_mu", C.c_float), ("clock_omega_relative_limit", C.c_float), ("costas_max_offset_hz", C.c_float),
        ("buffer_size", C.c_int), ("post_costas_dc", C.c_int),
        ("has_carrier", C.c_int), ("carrier_pll_bw", C.c_float), ("carrier_pll_max_offset", C.c_float), ("exact", C.c_int), ("chunk_len", C.c_int), ("warmup", C.c_int), ("device", C.c_int), ("freq_shift", C.c_double),
        ("doppler", C.c_int), ("doppler_alpha", C.c_float), ("custom_samplerate", C.c_double),
    ]


class DemodStats(C.Structure):
    _fields_ = [
        ("samples_in", C.c_uint64), ("symbols_out", C.c_uint64), ("freq_hz", C.c_float), ("final_sps", C.c_float),
        ("final_samplerate", C.c_float), ("buffer_size", C.c_int), ("resample_interp", C.c_int), ("resample_decim", C.c_int),
        ("chunks", C.c_uint32), ("chunks_fixed", C.c_uint32), ("chunks_rotated", C.c_uint32), ("chunks_inexact", C.c_uint32),
        ("chunks_forced", C.c_uint32),
    ]


class FecCfg(C.Structure):
    _fields_ = [
        ("decoder", C.c_int), ("constellation", C.c_int), ("iq_invert", C.c_int), ("cadu_size", C.c_int),
        ("viterbi_outsync_after", C.c_int), ("viterbi_ber_thresold", C.c_float), ("nrzm", C.c_int), ("derandomize", C.c_int),
        ("derand_after_rs", C.c_int), ("derand_start", C.c_int), ("rs_i", C.c_int), ("rs_fill_bytes", C.c_int),
        ("rs_dualbasis", C.c_int), ("rs_type", C.c_int), ("rs_usecheck", C.c_int), ("asm_sync", C.c_uint32),
        ("qpsk_swap_iq", C.c_int), ("qpsk_swap_diff", C.c_int), ("oqpsk_delay", C.c_int), ("oqpsk_method2", C.c_int), ("oqpsk_method3", C.c_int),
        ("conv_rate", C.c_int), ("device", C.c_int), ("invert_second_viterbi", C.c_int), ("m2x_interleaved", C.c_int),
    ]


class FecStats(C.Structure):
    _fields_ = [
        ("soft_in", C.c_uint64), ("blocks", C.c_uint64), ("bits_decoded", C.c_uint64), ("frames_deframed", C.c_uint64),
        ("frames_out", C.c_uint64), ("viterbi_ber", C.c_float), ("viterbi_lock", C.c_int), ("deframer_state", C.c_int),
        ("rs_errors", C.c_int * 8), ("vit_respec", C.c_uint32), ("tb_respec", C.c_uint32), ("viterbi2_ber", C.c_float), ("viterbi2_lock", C.c_int), ("watchdog_events", C.c_uint32),
    ]


class LdpcCfg(C.Structure):
    _fields_ = [("framesize", C.c_int), ("rate", C.c_int), ("batch", C.c_int), ("device", C.c_int)]


class BchCfg(C.Structure):
    _fields_ = [("framesize", C.c_int), ("rate", C.c_int), ("device", C.c_int)]


class LdpcInfo(C.Structure):
    _fields_ = [("code_len", C.c_int), ("data_len", C.c_int), ("layers", C.c_int), ("links_total", C.c_int), ("max_phases", C.c_int),
                ("layers_with_shared_bits", C.c_int), ("msg_bytes_per_frame", C.c_uint64)]


class Dvbs2Cfg(C.Structure):
    """sdhip_dvbs2_cfg (include/sdhip.h): DVBS2DemodModule's JSON keys."""
    _fields_ = [("front", DemodCfg), ("freq_prop_factor", C.c_float), ("modcod", C.c_int), ("shortframes", C.c_int), ("pilots", C.c_int), ("sof_thresold", C.c_float),
                ("ldpc_trials", C.c_int), ("ldpc_batch", C.c_int), ("lut_bits", C.c_void_p), ("lut_phase_error", C.c_void_p), ("lut_resolution", C.c_int)]


class Dvbs2Stats(C.Structure):
    _fields_ = [("samples_in", C.c_uint64), ("plframes", C.c_uint64), ("bbframes", C.c_uint64), ("snr", C.c_float), ("peak_snr", C.c_float), ("freq_hz", C.c_float),
                ("pll_freq", C.c_float), ("ldpc_trials", C.c_float), ("bch_corrections", C.c_float), ("detected_modcod", C.c_int), ("detected_shortframes", C.c_int),
                ("detected_pilots", C.c_int), ("pll_lanes", C.c_uint32), ("pll_rerun", C.c_uint32), ("pll_forced", C.c_uint32), ("pll_serial_frames", C.c_uint32), ("pll_branch_tries", C.c_uint32)]


class Vcdu(C.Structure):
    _fields_ = [("version", C.c_uint8), ("spacecraft_id", C.c_uint16), ("vcid", C.c_uint8), ("vcdu_counter", C.c_uint32), ("replay_flag", C.c_uint8)]


class AosPacket(C.Structure):
    _fields_ = [("header", C.c_uint8 * 6), ("version", C.c_uint8), ("type", C.c_uint8), ("secondary_header_flag", C.c_uint8), ("sequence_flag", C.c_uint8),
                ("apid", C.c_uint16), ("packet_sequence_count", C.c_uint16), ("packet_length", C.c_uint16), ("frame", C.c_uint32), ("payload_size", C.c_uint32),
                ("payload_offset", C.c_uint64)]


class LrptCfg(C.Structure):
    _fields_ = [("diff_decode", C.c_int), ("device", C.c_int)]


class LrptStats(C.Structure):
    _fields_ = [("soft_in", C.c_uint64), ("frames_seen", C.c_uint64), ("frames_out", C.c_uint64), ("viterbi_ber", C.c_float), ("correlator_lock", C.c_int),
                ("cor", C.c_int), ("rs_errors", C.c_int * 4)]


# dvbs2_code_rate_t (common/codings/dvb-s2/dvbs2.h:9-23)
S2_RATES = {"1/4": 0, "1/3": 1, "2/5": 2, "1/2": 3, "3/5": 4, "2/3": 5, "3/4": 6, "4/5": 7, "5/6": 8, "7/8": 9, "8/9": 10, "9/10": 11}


class NdspPskCfg(C.Structure):
    """sdhip_ndsp_psk_cfg (include/sdhip.h): the set_cfg() keys of satdump::ndsp::PSKDemodHierBlock."""
    _fields_ = [("device", C.c_int), ("constellation", C.c_int), ("samplerate", C.c_double), ("symbolrate", C.c_double),
                ("rrc_gain", C.c_double), ("rrc_alpha", C.c_double), ("rrc_ntaps", C.c_int),
                ("agc_rate", C.c_float), ("agc_reference", C.c_float), ("agc_gain", C.c_float), ("agc_max_gain", C.c_float),
                ("rec_omega", C.c_float), ("rec_omegaGain", C.c_float), ("rec_mu", C.c_float), ("rec_muGain", C.c_float), ("rec_omegaLimit", C.c_float),
                ("rec_nfilt", C.c_int), ("rec_ntaps", C.c_int), ("pll_loop_bw", C.c_float), ("pll_freq_limit", C.c_float),
                ("exact", C.c_int), ("chunk_len", C.c_int), ("warmup", C.c_int)]


class SdhipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdhipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                             "There is no CPU fallback.")
        if "libsdhip_emu" in os.path.basename(LIB_PATH) and os.environ.get("SDHIP_TESTING_TWIN") != "1":
            raise SdhipError("the host twin (tests/emu) is test infrastructure, not a backend: there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.sdhip_last_error.restype = C.c_char_p
        L.sdhip_version.restype = C.c_char_p
        L.sdhip_fec_create.restype = C.c_void_p
        L.sdhip_fec_create.argtypes = [C.POINTER(FecCfg)]
        L.sdhip_fec_destroy.argtypes = [C.c_void_p]
        L.sdhip_fec_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sdhip_fec_pull.restype = C.c_int64
        L.sdhip_fec_pull.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sdhip_fec_process_dev.restype = C.c_int64
        L.sdhip_fec_process_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.sdhip_fec_get_stats.argtypes = [C.c_void_p, C.POINTER(FecStats)]
        if hasattr(L, "sdhip_fec_flush"):
            L.sdhip_fec_flush.restype = C.c_int64
            L.sdhip_fec_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sdhip_fec_get_block_taps.restype = C.c_int64
        L.sdhip_fec_get_block_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.sdhip_fec_cfg_default.argtypes = [C.POINTER(FecCfg)]
        L.sdhip_op_ccdecoder.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sdhip_op_rs_decode.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        if hasattr(L, "sdhip_op_viterbi27"):
            L.sdhip_op_viterbi27.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        if hasattr(L, "sdhip_demod_create"):
            L.sdhip_demod_create.restype = C.c_void_p
            L.sdhip_demod_create.argtypes = [C.POINTER(DemodCfg)]
            L.sdhip_demod_destroy.argtypes = [C.c_void_p]
            L.sdhip_demod_cfg_default.argtypes = [C.POINTER(DemodCfg)]
            L.sdhip_demod_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            L.sdhip_demod_flush.argtypes = [C.c_void_p]
            L.sdhip_demod_pull.restype = C.c_int64
            L.sdhip_demod_pull.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            L.sdhip_demod_process_dev.restype = C.c_int64
            L.sdhip_demod_process_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
            L.sdhip_demod_get_stats.argtypes = [C.c_void_p, C.POINTER(DemodStats)]
            if hasattr(L, "sdhip_demod_doppler_targets"):
                L.sdhip_demod_doppler_targets.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            L.sdhip_op_block.restype = C.c_int64
            L.sdhip_op_block.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        if hasattr(L, "sdhip_dvbs2_front_create"):
            L.sdhip_dvbs2_front_create.restype = C.c_void_p
            L.sdhip_dvbs2_front_create.argtypes = [C.POINTER(DemodCfg)]
        if hasattr(L, "sdhip_ndsp_psk_demod_create"):
            L.sdhip_ndsp_psk_cfg_default.argtypes = [C.POINTER(NdspPskCfg)]
            L.sdhip_ndsp_psk_demod_create.restype = C.c_void_p
            L.sdhip_ndsp_psk_demod_create.argtypes = [C.POINTER(NdspPskCfg)]
            L.sdhip_ndsp_psk_demod_destroy.argtypes = [C.c_void_p]
            if hasattr(L, "sdhip_ndsp_block_create"):
                L.sdhip_ndsp_block_create.restype = C.c_void_p
                L.sdhip_ndsp_block_create.argtypes = [C.c_int, C.POINTER(NdspPskCfg)]
            L.sdhip_ndsp_psk_demod_work_dev.restype = C.c_int64
            L.sdhip_ndsp_psk_demod_work_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.sdhip_ndsp_psk_demod_work.restype = C.c_int64
            L.sdhip_ndsp_psk_demod_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.sdhip_ndsp_psk_demod_get_stats.argtypes = [C.c_void_p, C.POINTER(DemodStats)]
        if hasattr(L, "sdhip_ldpc_create"):
            L.sdhip_ldpc_create.restype = C.c_void_p
            L.sdhip_ldpc_create.argtypes = [C.POINTER(LdpcCfg)]
            L.sdhip_ldpc_destroy.argtypes = [C.c_void_p]
            L.sdhip_ldpc_get_info.argtypes = [C.c_void_p, C.POINTER(LdpcInfo)]
            L.sdhip_ldpc_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            L.sdhip_ldpc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        if hasattr(L, "sdhip_s2_ts_create"):  # dvbs2_ts_extractor (csrc/dvbs2_ts.hip)
            L.sdhip_s2_ts_create.restype = C.c_void_p
            L.sdhip_s2_ts_create.argtypes = [C.c_int, C.c_int]
            L.sdhip_s2_ts_destroy.argtypes = [C.c_void_p]
            L.sdhip_s2_ts_process.restype = C.c_int64
            L.sdhip_s2_ts_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
            L.sdhip_s2_ts_process_dev.restype = C.c_int64
            L.sdhip_s2_ts_process_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        if hasattr(L, "sdhip_bch_create"):
            L.sdhip_bch_create.restype = C.c_void_p
            L.sdhip_bch_create.argtypes = [C.POINTER(BchCfg)]
            L.sdhip_bch_destroy.argtypes = [C.c_void_p]
            L.sdhip_bch_dims.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.sdhip_bch_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            L.sdhip_bch_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            L.sdhip_s2_pack_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
            L.sdhip_bb_descramble_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            L.sdhip_s2_deinterleave_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(L, "sdhip_s2_bb_to_soft_dev"):
            L.sdhip_s2_bb_to_soft_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            L.sdhip_s2_cfg.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.sdhip_s2_pll_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
            L.sdhip_op_atan2f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            L.sdhip_s2_pl_sync_dev.restype = C.c_int64
            L.sdhip_s2_pl_sync_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        if hasattr(L, "sdhip_s2_pll_frames_dev"):
            L.sdhip_s2_pll_frames_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                  C.c_void_p]
        if hasattr(L, "sdhip_dvbs2_demod_create"):
            L.sdhip_dvbs2_cfg_default.argtypes = [C.POINTER(Dvbs2Cfg)]
            L.sdhip_dvbs2_demod_create.restype = C.c_void_p
            L.sdhip_dvbs2_demod_create.argtypes = [C.POINTER(Dvbs2Cfg)]
            L.sdhip_dvbs2_demod_destroy.argtypes = [C.c_void_p]
            L.sdhip_dvbs2_demod_bbframe_bytes.argtypes = [C.c_void_p]
            L.sdhip_dvbs2_demod_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            L.sdhip_dvbs2_demod_flush.argtypes = [C.c_void_p]
            L.sdhip_dvbs2_demod_pull.restype = C.c_int64
            L.sdhip_dvbs2_demod_pull.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            L.sdhip_dvbs2_demod_process_dev.restype = C.c_int64
            L.sdhip_dvbs2_demod_process_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
            L.sdhip_dvbs2_demod_symbols_dev.restype = C.c_int64
            L.sdhip_dvbs2_demod_symbols_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.sdhip_dvbs2_demod_get_stats.argtypes = [C.c_void_p, C.POINTER(Dvbs2Stats)]
        if hasattr(L, "sdhip_aos_demux_create"):
            L.sdhip_aos_parse_vcdu_dev.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            L.sdhip_aos_select_vcid_dev.restype = C.c_int64
            L.sdhip_aos_select_vcid_dev.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
            L.sdhip_aos_demux_create.restype = C.c_void_p
            L.sdhip_aos_demux_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
            L.sdhip_aos_demux_destroy.argtypes = [C.c_void_p]
            L.sdhip_aos_demux_work_dev.restype = C.c_int64
            L.sdhip_aos_demux_work_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        if hasattr(L, "sdhip_lrpt_create"):
            L.sdhip_lrpt_cfg_default.argtypes = [C.POINTER(LrptCfg)]
            L.sdhip_lrpt_create.restype = C.c_void_p
            L.sdhip_lrpt_create.argtypes = [C.POINTER(LrptCfg)]
            L.sdhip_lrpt_destroy.argtypes = [C.c_void_p]
            L.sdhip_lrpt_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            L.sdhip_lrpt_pull.restype = C.c_int64
            L.sdhip_lrpt_pull.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            L.sdhip_lrpt_process_dev.restype = C.c_int64
            L.sdhip_lrpt_process_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.sdhip_lrpt_get_stats.argtypes = [C.c_void_p, C.POINTER(LrptStats)]
        L.sdhip_prof_enable.argtypes = [C.c_int]
        L.sdhip_pool_enable.argtypes = [C.c_int]
        L.sdhip_prof_get.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        _lib = L
    return _lib


def pool_enable(on: bool = True):
    """Park the device / pinned blocks of destroyed handles for the next handles (sdhip_pool_enable)."""
    lib().sdhip_pool_enable(int(on))


def prof_enable(on: bool = True):
    lib().sdhip_prof_enable(int(on))


def prof_reset():
    lib().sdhip_prof_reset()


def prof_get() -> dict:
    """{kernel name: (total ms, launches)} measured with HIP events on the launch stream (sdhip_prof_get)."""
    L = lib()
    n = L.sdhip_prof_get(-1, None, 0, None, None)
    out = {}
    for i in range(n):
        buf = C.create_string_buffer(128)
        ms, cnt = C.c_double(0), C.c_longlong(0)
        L.sdhip_prof_get(i, buf, 128, C.byref(ms), C.byref(cnt))
        out[buf.value.decode()] = (ms.value, cnt.value)
    return out


def last_error() -> str:
    return lib().sdhip_last_error().decode()


def _check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise SdhipError(f"{what} failed: {last_error()}")
    return rc


def fec_cfg(**kw) -> FecCfg:
    c = FecCfg()
    lib().sdhip_fec_cfg_default(C.byref(c))
    for k, v in kw.items():
        if k == "constellation" and isinstance(v, str):
            v = CONSTELLATIONS[v]
        setattr(c, k, v)
    return c


def demod_cfg(**kw) -> DemodCfg:
    c = DemodCfg()
    lib().sdhip_demod_cfg_default(C.byref(c))
    for k, v in kw.items():
        if k == "constellation" and isinstance(v, str):
            v = CONSTELLATIONS[v]
        setattr(c, k, v)
    return c


class FecDecoder:
    """ccsds_conv_concat_decoder / metop_ahrpt_decoder on one GPU stream (one handle = one stream)."""

    def __init__(self, cfg: FecCfg):
        self.cfg = cfg
        self.h = lib().sdhip_fec_create(C.byref(cfg))
        if not self.h:
            raise SdhipError(f"sdhip_fec_create failed: {last_error()}")
        self.cadu_bytes = 1024 if cfg.decoder in (DEC_METOP_AHRPT, DEC_FENGYUN_AHRPT, DEC_FENGYUN_MPT) else cfg.cadu_size // 8

    def close(self):
        if self.h:
            lib().sdhip_fec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, soft: np.ndarray):
        s = np.ascontiguousarray(soft, dtype=np.int8)
        _check(lib().sdhip_fec_push(self.h, s.ctypes.data_as(C.c_void_p), s.size), "sdhip_fec_push")

    def pull(self, max_frames: int = 1 << 20) -> np.ndarray:
        out = np.zeros((max_frames, self.cadu_bytes), dtype=np.uint8)
        n = _check(lib().sdhip_fec_pull(self.h, out.ctypes.data_as(C.c_void_p), max_frames), "sdhip_fec_pull")
        return out[:n].copy()

    def process_dev(self, soft_ptr: int, n: int, cadu_ptr: int, cap_frames: int) -> int:
        return _check(lib().sdhip_fec_process_dev(self.h, C.c_void_p(soft_ptr), n, C.c_void_p(cadu_ptr), cap_frames), "sdhip_fec_process_dev")

    def flush(self, cadu_ptr: int = 0, cap_frames: int = 0) -> int:
        """end of the input (m2x_interleaved handles: the module's last reads); frames to cadu_ptr (device) or, with 0, to the queue pull() reads"""
        return _check(lib().sdhip_fec_flush(self.h, C.c_void_p(cadu_ptr), cap_frames), "sdhip_fec_flush")

    def stats(self) -> FecStats:
        st = FecStats()
        lib().sdhip_fec_get_stats(self.h, C.byref(st))
        return st

    def block_taps(self):
        n = lib().sdhip_fec_get_block_taps(self.h, None, None, 0)
        ber = np.zeros(n, dtype=np.float32)
        st = np.zeros(n, dtype=np.int32)
        lib().sdhip_fec_get_block_taps(self.h, ber.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), n)
        return ber, st


class PskDemod:
    """psk_demod on one GPU stream. front_only=True: the DVB-S2 demodulator's front end (no Costas loop; take the symbols, sdhip_dvbs2_front_create)."""

    def __init__(self, cfg: DemodCfg, front_only: bool = False):
        self.cfg = cfg
        self.h = (lib().sdhip_dvbs2_front_create if front_only else lib().sdhip_demod_create)(C.byref(cfg))
        if not self.h:
            raise SdhipError(f"sdhip_demod_create failed: {last_error()}")

    def close(self):
        if self.h:
            lib().sdhip_demod_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, iq: np.ndarray, fmt: int = FMT_CF32):
        a = np.ascontiguousarray(iq)
        nsamp = a.size if a.dtype == np.complex64 else a.size // 2
        _check(lib().sdhip_demod_push(self.h, a.ctypes.data_as(C.c_void_p), nsamp, fmt), "sdhip_demod_push")

    def flush(self):
        _check(lib().sdhip_demod_flush(self.h), "sdhip_demod_flush")

    def pull(self, cap: int = 1 << 26, out: np.ndarray | None = None) -> np.ndarray:
        """Up to cap soft bytes; with `out` (an int8 array the caller owns) nothing is allocated and a view of it is returned."""
        if out is None:
            out = np.empty(cap, dtype=np.int8)
            n = _check(lib().sdhip_demod_pull(self.h, out.ctypes.data_as(C.c_void_p), cap), "sdhip_demod_pull")
            return out[:n].copy()
        n = _check(lib().sdhip_demod_pull(self.h, out.ctypes.data_as(C.c_void_p), min(cap, out.size)), "sdhip_demod_pull")
        return out[:n]

    def process_dev(self, iq_ptr: int, nsamples: int, fmt: int, soft_ptr: int, soft_cap: int, syms_ptr: int = 0, syms_cap: int = 0, final: bool = True) -> int:
        return _check(lib().sdhip_demod_process_dev(self.h, C.c_void_p(iq_ptr), nsamples, fmt, C.c_void_p(soft_ptr), soft_cap,
                                                    C.c_void_p(syms_ptr) if syms_ptr else None, syms_cap, int(final)), "sdhip_demod_process_dev")

    def stats(self) -> DemodStats:
        st = DemodStats()
        lib().sdhip_demod_get_stats(self.h, C.byref(st))
        return st

    def set_tap(self, mode: int):
        """Test tap (sdhip_demod_set_tap): mode 1 = d_syms carries the clock recovery's int64 grid position of every symbol instead of the symbol."""
        L = lib()
        L.sdhip_demod_set_tap.argtypes = [C.c_void_p, C.c_int]
        _check(L.sdhip_demod_set_tap(self.h, int(mode)), "sdhip_demod_set_tap")

    def doppler_targets(self, targets):
        """The Doppler rotator's target frequencies (rad / sample) for the source buffers to come (sdhip_demod_doppler_targets)."""
        t = np.ascontiguousarray(targets, dtype=np.float32)
        _check(lib().sdhip_demod_doppler_targets(self.h, t.ctypes.data_as(C.c_void_p), len(t)), "sdhip_demod_doppler_targets")


class LdpcDecoder:
    """dvbs2::BBFrameLDPC's decode on the GPU (include/sdhip.h, sdhip_ldpc_*): framesize 0 normal / 1 short, rate a key of S2_RATES or
    the enum value, batch = frames per reference decode call (its SIMD width)."""

    def __init__(self, framesize=0, rate="2/3", batch=1, device=0):
        cfg = LdpcCfg(int(framesize), S2_RATES[rate] if isinstance(rate, str) else int(rate), int(batch), int(device))
        self.batch = int(batch)
        self.h = lib().sdhip_ldpc_create(C.byref(cfg))
        if not self.h:
            raise SdhipError(lib().sdhip_last_error().decode())
        self.info = LdpcInfo()
        lib().sdhip_ldpc_get_info(self.h, C.byref(self.info))

    def close(self):
        if self.h:
            lib().sdhip_ldpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, frames: np.ndarray, max_trials: int = 25):
        """frames: int8 [nframes, code_len] (host), decoded in place -> trials per batch (update passes run, -1 = not converged)."""
        assert frames.dtype == np.int8 and frames.flags.c_contiguous and frames.shape[1] == self.info.code_len
        tr = np.zeros(frames.shape[0] // self.batch, dtype=np.int32)
        r = lib().sdhip_ldpc_decode(self.h, frames.ctypes.data_as(C.c_void_p), frames.shape[0], max_trials, tr.ctypes.data_as(C.c_void_p))
        if r < 0:
            raise SdhipError(lib().sdhip_last_error().decode())
        return tr

    def decode_dev(self, d_frames_ptr: int, nframes: int, max_trials: int, d_trials_ptr: int) -> int:
        r = lib().sdhip_ldpc_decode_dev(self.h, d_frames_ptr, nframes, max_trials, d_trials_ptr)
        if r < 0:
            raise SdhipError(lib().sdhip_last_error().decode())
        return r


class BchDecoder:
    """dvbs2::BBFrameBCH::decode on the GPU (include/sdhip.h, sdhip_bch_*)."""

    def __init__(self, framesize=0, rate="2/3", device=0):
        cfg = BchCfg(int(framesize), S2_RATES[rate] if isinstance(rate, str) else int(rate), int(device))
        self.h = lib().sdhip_bch_create(C.byref(cfg))
        if not self.h:
            raise SdhipError(lib().sdhip_last_error().decode())
        k, n = C.c_int(), C.c_int()
        lib().sdhip_bch_dims(self.h, C.byref(k), C.byref(n))
        self.kbch, self.nbch = k.value, n.value

    def close(self):
        if self.h:
            lib().sdhip_bch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, frames: np.ndarray):
        """frames uint8 [nframes, stride >= nbch / 8] (host), corrected in place -> corrections per frame."""
        assert frames.dtype == np.uint8 and frames.flags.c_contiguous
        corr = np.zeros(frames.shape[0], dtype=np.int32)
        r = lib().sdhip_bch_decode(self.h, frames.ctypes.data_as(C.c_void_p), frames.shape[0], frames.shape[1], corr.ctypes.data_as(C.c_void_p))
        if r < 0:
            raise SdhipError(lib().sdhip_last_error().decode())
        return corr

    def decode_dev(self, d_frames_ptr, nframes, stride, d_corr_ptr):
        if lib().sdhip_bch_decode_dev(self.h, d_frames_ptr, nframes, stride, d_corr_ptr) < 0:
            raise SdhipError(lib().sdhip_last_error().decode())

    def descramble_dev(self, d_frames_ptr, nframes, stride):
        if lib().sdhip_bb_descramble_dev(self.h, d_frames_ptr, nframes, stride) < 0:
            raise SdhipError(lib().sdhip_last_error().decode())

    def pack_dev(self, d_soft_ptr, soft_stride, nframes, d_out_ptr, out_stride):
        if lib().sdhip_s2_pack_dev(self.h, d_soft_ptr, soft_stride, nframes, d_out_ptr, out_stride) < 0:
            raise SdhipError(lib().sdhip_last_error().decode())
