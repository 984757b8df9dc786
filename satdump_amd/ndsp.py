"""Host-side mirror of the reference's ndsp PSK demodulator, satdump::ndsp::PSKDemodHierBlock (src-core/dsp/hier/psk_demod.h:22-249),
over the C ABI (include/sdhip.h, sdhip_ndsp_psk_demod_*): same block id, same set_cfg() / get_cfg() keys and result codes, one work()
per DSPBuffer. The arithmetic runs in libsdhip.so on the GPU; there is no CPU path here."""
import ctypes as C

import numpy as np

from . import capi

# Block::cfg_res_t (src-core/dsp/block.h:258-264)
RES_OK, RES_LISTUPD, RES_IOUPD, RES_ERR = 0, 1, 2, 3

_CONST = {"bpsk": capi.BPSK, "qpsk": capi.QPSK}
# hier key -> cfg field (psk_demod.h:230-249 strips the rrc_/agc_/rec_/pll_ prefix and hands the rest to the member block)
_ADVANCED = {
    "rrc_gain": "rrc_gain", "rrc_alpha": "rrc_alpha", "rrc_ntaps": "rrc_ntaps",
    "agc_rate": "agc_rate", "agc_reference": "agc_reference", "agc_gain": "agc_gain", "agc_max_gain": "agc_max_gain",
    "rec_omega": "rec_omega", "rec_omegaGain": "rec_omegaGain", "rec_mu": "rec_mu", "rec_muGain": "rec_muGain", "rec_omegaLimit": "rec_omegaLimit",
    "rec_nfilt": "rec_nfilt", "rec_ntaps": "rec_ntaps",
    "pll_loop_bw": "pll_loop_bw", "pll_freq_limit": "pll_freq_limit",
}


class PSKDemodHierBlock:
    """psk_demod_cc. The reference applies a changed parameter by re-initialising the member block it belongs to (the AGC on any agc_ key,
    the clock recovery at its next buffer, the filter at its next buffer, the loop on loop_bw / freq_limit); here the engine is rebuilt at
    the next work() after a set_cfg(), which is the same thing when the block is configured before the stream starts -- the only use
    the reference makes of it (pipeline/modules/demod/module_demod_ndsp.cpp:22-24)."""

    d_id = "psk_demod_cc"

    def __init__(self, device: int = 0, exact: bool = False, capi_mod=None):
        self._capi = capi_mod or capi  # (the test suite's host twin hands in its own binding)
        self._cfg = self._capi.NdspPskCfg()
        self._capi.lib().sdhip_ndsp_psk_cfg_default(C.byref(self._cfg))
        self._cfg.device = device
        self._cfg.exact = int(exact)
        self._constellation = "bpsk"
        self._advanced = False
        self._h = None

    # ---- configuration
    def get_cfg_list(self):
        keys = ["constellation", "samplerate", "symbolrate", "advanced", "pll_freq", "snr"]
        return keys + (list(_ADVANCED) if self._advanced else [])

    def set_cfg(self, key: str, v) -> int:
        if key == "constellation":
            if v not in _CONST:  # psk_demod.h:205: anything but bpsk / qpsk falls through to RES_ERR
                return RES_ERR
            self._constellation = v
            self._cfg.constellation = _CONST[v]
        elif key in ("samplerate", "symbolrate"):
            setattr(self._cfg, key, float(v))
        elif key == "advanced":
            self._advanced = bool(v)
            return RES_LISTUPD
        elif key in _ADVANCED:
            f = _ADVANCED[key]
            setattr(self._cfg, f, type(getattr(self._cfg, f))(v))
        else:
            return RES_ERR
        self._drop()
        return RES_OK

    def get_cfg(self, key: str):
        if key == "constellation":
            return self._constellation
        if key in ("samplerate", "symbolrate"):
            return getattr(self._cfg, key)
        if key == "advanced":
            return self._advanced
        if key == "pll_freq":  # rad_to_hz(pll freq, symbolrate), psk_demod.h:170
            return float(self.stats().freq_hz) if self._h else 0.0
        if key in _ADVANCED:
            return getattr(self._cfg, _ADVANCED[key])
        return None

    # ---- stream
    def _handle(self):
        if self._h is None:
            self._h = self._capi.lib().sdhip_ndsp_psk_demod_create(C.byref(self._cfg))
            if not self._h:
                raise self._capi.SdhipError(f"sdhip_ndsp_psk_demod_create failed: {self._capi.last_error()}")
        return self._h

    def _drop(self):
        if self._h:
            self._capi.lib().sdhip_ndsp_psk_demod_destroy(self._h)
            self._h = None

    def start(self):
        self._drop()
        self._handle()

    def stop(self, stop_now: bool = False, force: bool = False):
        self._drop()

    def work(self, samples: np.ndarray) -> np.ndarray:
        """One input buffer (complex64) -> the symbols it produces (complex64)."""
        x = np.ascontiguousarray(samples, dtype=np.complex64)
        out = np.zeros(len(x) + 64, dtype=np.complex64)
        n = self._capi._check(self._capi.lib().sdhip_ndsp_psk_demod_work(self._handle(), x.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p), len(out)),
                        "sdhip_ndsp_psk_demod_work")
        return out[:n].copy()

    def work_dev(self, d_in_ptr: int, nsamples: int, d_out_ptr: int, out_cap: int) -> int:
        return self._capi._check(self._capi.lib().sdhip_ndsp_psk_demod_work_dev(self._handle(), C.c_void_p(d_in_ptr), nsamples, C.c_void_p(d_out_ptr), out_cap),
                           "sdhip_ndsp_psk_demod_work_dev")

    def stats(self) -> capi.DemodStats:
        st = self._capi.DemodStats()
        self._capi.lib().sdhip_ndsp_psk_demod_get_stats(self._handle(), C.byref(st))
        return st

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass


# ---- the chain's member blocks as flowgraph nodes of their own (include/sdhip.h, sdhip_ndsp_block_create)
NDSP_HIER, NDSP_RRC_FIR, NDSP_AGC, NDSP_MM, NDSP_COSTAS, NDSP_GARDNER, NDSP_AGC_FAST, NDSP_COSTAS_FAST, NDSP_MM_FAST = 0, 1, 2, 3, 4, 5, 6, 7, 8
_SINGLE = {
    # block id -> (kind, {block key: cfg field})   (the keys of dsp/agc/agc.h:38-78, dsp/filter/rrc.h:34-66, dsp/clock_recovery/clock_recovery_mm.h:70-130, dsp/pll/costas.h:55-90)
    "agc_cc": (NDSP_AGC, {"rate": "agc_rate", "reference": "agc_reference", "gain": "agc_gain", "max_gain": "agc_max_gain"}),
    "agc_fast_cc": (NDSP_AGC_FAST, {"rate": "agc_rate", "reference": "agc_reference", "gain": "agc_gain", "max_gain": "agc_max_gain"}),  # dsp/agc/agc_fast.h:45-84
    "rrc_fir_cc": (NDSP_RRC_FIR, {"gain": "rrc_gain", "samplerate": "samplerate", "symbolrate": "symbolrate", "alpha": "rrc_alpha", "ntaps": "rrc_ntaps"}),
    "clock_recovery_mm_cc": (NDSP_MM, {"omega": "rec_omega", "omegaGain": "rec_omegaGain", "mu": "rec_mu", "muGain": "rec_muGain", "omegaLimit": "rec_omegaLimit",
                                       "nfilt": "rec_nfilt", "ntaps": "rec_ntaps"}),
    "costas_cc": (NDSP_COSTAS, {"loop_bw": "pll_loop_bw", "freq_limit": "pll_freq_limit"}),
    # dsp/clock_recovery/clock_recovery_gardner.h:15-21, 57-130: the M&M block's keys (its own default omega is 0: set it)
    "clock_recovery_gardner_cc": (NDSP_GARDNER, {"omega": "rec_omega", "omegaGain": "rec_omegaGain", "mu": "rec_mu", "muGain": "rec_muGain", "omegaLimit": "rec_omegaLimit",
                                                 "nfilt": "rec_nfilt", "ntaps": "rec_ntaps"}),
    # dsp/pll/costas_fast.h:63-104 (the Costas block's keys) and dsp/clock_recovery/clock_recovery_mm_fast.h:59-116 (the M&M block's without the bank's shape). Both run
    # lane-per-chunk with a bit-exact hand-off (the clock recovery under all five cadences of its rate update per chunk; include/sdhip.h): bit for bit the blocks
    "costas_fast_cc": (NDSP_COSTAS_FAST, {"loop_bw": "pll_loop_bw", "freq_limit": "pll_freq_limit"}),
    "fast_clock_recovery_mm_cc": (NDSP_MM_FAST, {"omega": "rec_omega", "omegaGain": "rec_omegaGain", "mu": "rec_mu", "muGain": "rec_muGain", "omegaLimit": "rec_omegaLimit"}),
}
_ORDER = {2: capi.BPSK, 4: capi.QPSK, 8: capi.PSK8}


class SingleBlock:
    """agc_cc / rrc_fir_cc / clock_recovery_mm_cc / costas_cc / clock_recovery_gardner_cc: one member block of the chain with the reference block's own keys and defaults
    (AGCBlock: reference 1.0 -- the hier block is what sets 0.6 --, MMClockRecoveryBlock: omega 2, CostasBlock: order 2), state carried across work() calls."""

    def __init__(self, block_id: str, device: int = 0, exact: bool = False, capi_mod=None):
        if block_id not in _SINGLE:
            raise ValueError("unknown block " + block_id)
        self.d_id = block_id
        self._capi = capi_mod or capi
        self._kind, self._keys = _SINGLE[block_id]
        self._cfg = self._capi.NdspPskCfg()
        self._capi.lib().sdhip_ndsp_psk_cfg_default(C.byref(self._cfg))
        self._cfg.device, self._cfg.exact = device, int(exact)
        self._cfg.agc_reference = 1.0      # agc.h:15
        self._cfg.rec_omega = 2.0          # clock_recovery_mm.h:17
        self._cfg.samplerate, self._cfg.symbolrate = 6e6, 2e6
        self._h = None

    def set_cfg(self, key: str, v) -> int:
        if self._kind in (NDSP_COSTAS, NDSP_COSTAS_FAST) and key == "order":
            if int(v) not in _ORDER:
                return RES_ERR
            self._cfg.constellation = _ORDER[int(v)]
        elif key in self._keys:
            f = self._keys[key]
            setattr(self._cfg, f, type(getattr(self._cfg, f))(v))
        else:
            return RES_ERR
        self._drop()
        return RES_OK

    def _handle(self):
        if self._h is None:
            self._h = self._capi.lib().sdhip_ndsp_block_create(self._kind, C.byref(self._cfg))
            if not self._h:
                raise self._capi.SdhipError(f"sdhip_ndsp_block_create failed: {self._capi.last_error()}")
        return self._h

    def _drop(self):
        if self._h:
            self._capi.lib().sdhip_ndsp_psk_demod_destroy(self._h)
            self._h = None

    def work(self, samples: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(samples, dtype=np.complex64)
        out = np.zeros(len(x) + 64, dtype=np.complex64)
        n = self._capi._check(self._capi.lib().sdhip_ndsp_psk_demod_work(self._handle(), x.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p), len(out)),
                              "sdhip_ndsp_psk_demod_work")
        return out[:n].copy()

    def work_dev(self, d_in_ptr: int, nsamples: int, d_out_ptr: int, out_cap: int) -> int:
        return self._capi._check(self._capi.lib().sdhip_ndsp_psk_demod_work_dev(self._handle(), C.c_void_p(d_in_ptr), nsamples, C.c_void_p(d_out_ptr), out_cap),
                                 "sdhip_ndsp_psk_demod_work_dev")

    def stats(self) -> capi.DemodStats:
        st = self._capi.DemodStats()
        self._capi.lib().sdhip_ndsp_psk_demod_get_stats(self._handle(), C.byref(st))
        return st

    def stop(self, stop_now: bool = False, force: bool = False):
        self._drop()

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass
