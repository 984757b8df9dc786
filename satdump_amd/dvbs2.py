"""Python face of the DVB-S2 demodulator handle in libsdhip.so (include/sdhip.h, sdhip_dvbs2_demod_*; the engine is satdump_amd/csrc/dvbs2_engine.hip):
the module's parameter keys, defaults and error messages (satdump::pipeline::dvb::DVBS2DemodModule, plugins/dvb_support/dvbs2/
module_dvbs2_demod.{h,cpp}), baseband samples in, BBFRAME bytes out (`bch_decoder->dataSize() / 8` bytes per frame, what the module writes to its
.bbframe file). Everything -- the stage chain, the carry-over between calls (PL synchroniser ring, PLL state, frames waiting for a decoder group),
the statistics -- lives behind the C ABI since round 4; this file only parses the dictionary and moves buffers. There is no CPU path here.

freq_prop_factor: the module's frequency feedback runs on thread timing in the reference; the engine applies it at call boundaries in closed form
(dvbs2_engine.hip's header). exact=True (the serial schedules, bit for bit the reference's blocks chained) requires freq_prop_factor 0.

Device memory comes from a small adapter so that the same code runs on the GPU (torch) and on the test suite's host twin (numpy):
    mem.from_host(array) -> handle, mem.alloc(n, dtype) -> handle, mem.ptr(handle) -> int, mem.to_host(handle, n) -> np.ndarray, mem.device_index -> int."""
import ctypes as C

import numpy as np

from . import capi as _capi


class TorchMem:
    """Device buffers through torch (the GPU path)."""

    def __init__(self, device="cuda"):
        import torch
        self.t = torch
        self.device = torch.device(device)
        self.device_index = self.device.index if self.device.index is not None else (torch.cuda.current_device() if self.device.type == "cuda" else 0)
        self._dt = {np.int8: torch.int8, np.uint8: torch.uint8, np.int32: torch.int32, np.float32: torch.float32}

    def alloc(self, n, dtype):
        return self.t.zeros(int(n), dtype=self._dt[dtype], device=self.device)

    def from_host(self, a):
        return self.t.from_numpy(np.ascontiguousarray(a)).to(self.device)

    @staticmethod
    def ptr(h):
        return h.data_ptr()

    @staticmethod
    def to_host(h, n=None):
        return (h if n is None else h[:n]).cpu().numpy()


def parse_parameters(parameters: dict, lut_bits: np.ndarray, lut_phase_error: np.ndarray, capi=None, exact: bool = False, batch: int = 1, device: int = 0):
    """The module's constructor (module_dvbs2_demod.cpp:13-83 on module_demod_base.cpp:12-57): dictionary -> sdhip_dvbs2_cfg. The returned
    tuple keeps the table arrays alive for the create call."""
    capi = capi or _capi
    p = dict(parameters)
    if "samplerate" not in p:
        raise ValueError("Samplerate parameter must be present!")          # module_demod_base.cpp:18
    if "rrc_alpha" not in p:
        raise ValueError("RRC Alpha parameter must be present!")           # module_dvbs2_demod.cpp:22
    if "pll_bw" not in p:
        raise ValueError("PLL BW parameter must be present!")              # :30
    if "modcod" not in p:
        raise ValueError("MODCOD parameter must be present!")              # :58
    cfg = capi.Dvbs2Cfg()
    capi.lib().sdhip_dvbs2_cfg_default(C.byref(cfg))
    f = cfg.front
    f.samplerate, f.symbolrate = float(p["samplerate"]), float(p["symbolrate"])
    f.rrc_alpha, f.rrc_taps, f.pll_bw = float(p["rrc_alpha"]), int(p.get("rrc_taps", 31)), float(p["pll_bw"])
    f.agc_rate = float(p.get("agc_rate", 1e-2))
    if "clock_alpha" in p:                                                  # :35-40
        ca = np.float32(p["clock_alpha"])
        f.clock_gain_omega, f.clock_gain_mu = float(np.float32(float(ca) ** 2 / 4.0)), float(ca)
    for k in ("clock_gain_omega", "clock_mu", "clock_gain_mu", "clock_omega_relative_limit", "min_sps", "max_sps"):
        if k in p:
            setattr(f, k, float(p[k]))
    for k in ("dc_block", "iq_swap", "buffer_size", "chunk_len"):
        if k in p:
            setattr(f, k, int(p[k]))
    if "freq_shift" in p:
        f.freq_shift = float(p["freq_shift"])
    f.exact, f.device = int(exact), int(device)
    cfg.freq_prop_factor = float(p.get("freq_prop_factor", 0.01))
    cfg.modcod, cfg.shortframes, cfg.pilots = int(p["modcod"]), int(bool(p.get("shortframes", False))), int(bool(p.get("pilots", False)))
    cfg.sof_thresold = float(p.get("sof_thresold", 0.6))
    cfg.ldpc_trials = int(p.get("ldpc_trials", 10))
    cfg.ldpc_batch = int(batch)
    lb = np.ascontiguousarray(lut_bits, dtype=np.int8)
    lp = np.ascontiguousarray(lut_phase_error, dtype=np.float32)
    assert lb.ndim == 3 and lb.shape[0] == lb.shape[1] and lp.shape == lb.shape[:2]
    cfg.lut_bits, cfg.lut_phase_error, cfg.lut_resolution = lb.ctypes.data, lp.ctypes.data, lb.shape[0]
    return cfg, (lb, lp)


class DVBS2Demod:
    """dvbs2_demod. parameters: the module's JSON keys (samplerate, symbolrate, rrc_alpha, rrc_taps, pll_bw, clock_alpha / clock_gain_omega /
    clock_mu / clock_gain_mu / clock_omega_relative_limit, modcod, shortframes, pilots, sof_thresold, ldpc_trials, agc_rate, freq_prop_factor)
    plus lut_bits / lut_phase_error: the demapper table constellation_t::make_lut(256) builds on the host (its bits [256][256][bits] int8 and
    phase errors [256][256] float32) -- data the reference computes with the host libm; hand over what the reference built (tests: the
    compiled reference's) or any table of that layout."""

    def __init__(self, parameters: dict, lut_bits: np.ndarray, lut_phase_error: np.ndarray, mem=None, capi=None, exact: bool = False, batch: int = 1):
        self.capi = capi or _capi
        self.mem = mem or TorchMem()
        cfg, keep = parse_parameters(parameters, lut_bits, lut_phase_error, capi=self.capi, exact=exact, batch=batch, device=int(getattr(self.mem, "device_index", 0)))
        L = self.capi.lib()
        self.h = L.sdhip_dvbs2_demod_create(C.byref(cfg))
        del keep
        if not self.h:
            msg = self.capi.last_error()
            raise (ValueError if ("MODCOD" in msg or "32APSK" in msg) else self.capi.SdhipError)(msg)  # get_dvbs2_cfg's messages
        self.bbframe_bytes = int(L.sdhip_dvbs2_demod_bbframe_bytes(self.h))
        self.samples_per_frame_hint = 1

    def close(self):
        if getattr(self, "h", None):
            self.capi.lib().sdhip_dvbs2_demod_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stats(self) -> dict:
        st = self.capi.Dvbs2Stats()
        self.capi.lib().sdhip_dvbs2_demod_get_stats(self.h, C.byref(st))
        d = {k: getattr(st, k) for k, _ in st._fields_}
        d["frames"] = d["bbframes"]
        d["pls"] = None if st.detected_modcod < 0 else (st.detected_modcod << 2) | (st.detected_shortframes << 1) | st.detected_pilots
        d["freq"] = d["pll_freq"]
        return d

    def process_dev(self, iq_ptr: int, nsamples: int, fmt: int, out_ptr: int, cap_frames: int) -> int:
        n = self.capi.lib().sdhip_dvbs2_demod_process_dev(self.h, C.c_void_p(iq_ptr), int(nsamples), int(fmt), C.c_void_p(out_ptr), int(cap_frames))
        if n < 0:
            raise self.capi.SdhipError(self.capi.last_error())
        return int(n)

    def symbols_dev(self, syms_ptr: int, nsyms: int, out_ptr: int, cap_frames: int) -> int:
        n = self.capi.lib().sdhip_dvbs2_demod_symbols_dev(self.h, C.c_void_p(syms_ptr), int(nsyms), C.c_void_p(out_ptr), int(cap_frames))
        if n < 0:
            raise self.capi.SdhipError(self.capi.last_error())
        return int(n)

    def process(self, iq: np.ndarray, fmt=None) -> np.ndarray:
        """One batch of baseband samples (complex64, or the integer formats with fmt) -> the BBFRAMEs it completes, uint8 [nframes, bbframe_bytes]."""
        m, cap = self.mem, self.capi
        x = np.ascontiguousarray(iq)
        fmt = cap.FMT_CF32 if fmt is None else fmt
        nsamp = len(x) if x.dtype == np.complex64 else x.size // 2
        if nsamp == 0:
            return np.zeros((0, self.bbframe_bytes), dtype=np.uint8)
        d_x = m.from_host(x.view(np.float32) if x.dtype == np.complex64 else x)
        # frames a call can complete: what was carried (at most one decoder group and a frame of symbols) + what the samples hold
        cap_frames = nsamp // 4000 + 80
        d_out = m.alloc(cap_frames * self.bbframe_bytes, np.uint8)
        n = self.process_dev(m.ptr(d_x), nsamp, fmt, m.ptr(d_out), cap_frames)
        return m.to_host(d_out, n * self.bbframe_bytes).reshape(n, self.bbframe_bytes).copy()

    def push(self, iq: np.ndarray, fmt=None):
        x = np.ascontiguousarray(iq)
        fmt = self.capi.FMT_CF32 if fmt is None else fmt
        nsamp = len(x) if x.dtype == np.complex64 else x.size // 2
        if self.capi.lib().sdhip_dvbs2_demod_push(self.h, x.ctypes.data_as(C.c_void_p), nsamp, fmt) != 0:
            raise self.capi.SdhipError(self.capi.last_error())

    def flush(self):
        if self.capi.lib().sdhip_dvbs2_demod_flush(self.h) != 0:
            raise self.capi.SdhipError(self.capi.last_error())

    def pull(self, cap_frames: int = 4096) -> np.ndarray:
        out = np.zeros((cap_frames, self.bbframe_bytes), dtype=np.uint8)
        n = self.capi.lib().sdhip_dvbs2_demod_pull(self.h, out.ctypes.data_as(C.c_void_p), cap_frames)
        if n < 0:
            raise self.capi.SdhipError(self.capi.last_error())
        return out[:n].copy()
