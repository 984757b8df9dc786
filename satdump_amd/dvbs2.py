"""Host-side mirror of the reference's DVB-S2 demodulator module, satdump::pipeline::dvb::DVBS2DemodModule
(plugins/dvb_support/dvbs2/module_dvbs2_demod.{h,cpp}), over the C ABI (include/sdhip.h): the same parameter keys, defaults and error messages,
baseband samples in, BBFRAME bytes out (`bch_decoder->dataSize() / 8` bytes per frame, what the module writes to its .bbframe file). Every
stage is an entry point of libsdhip.so -- front end (AGC, RRC filter, clock recovery), PL synchroniser, frame PLL, soft demapper stage, LDPC,
repack, BCH, BB descrambler --; this file is plumbing: buffers, the carry-over between calls (unconsumed symbols as the PL synchroniser's
ring buffer keeps them, the PLL state, frames waiting for a full LDPC batch). There is no CPU path here.

One thing the module does that is NOT mirrored: it feeds the PLL's frequency back into a rotator in front of the PL synchroniser
(`freq_prop_factor`, default 0.01, module_dvbs2_demod.cpp:204-206) from another thread, whenever a frame happens to come out -- its output
depends on thread timing and cannot be reproduced bit for bit by anything. `freq_prop_factor` must be 0 here (the frame PLL then tracks the
whole offset itself, as it does in the reference with that setting).

Device memory comes from a small adapter so that the same code runs on the GPU (torch) and on the test suite's host twin (numpy):
    mem.alloc(n, dtype) -> handle, mem.ptr(handle) -> int, mem.to_host(handle, n) -> np.ndarray, mem.from_host(array) -> handle."""
import ctypes as C

import numpy as np

from . import capi as _capi


class TorchMem:
    """Device buffers through torch (the GPU path)."""

    def __init__(self, device="cuda"):
        import torch
        self.t = torch
        self.device = device
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


class DVBS2Demod:
    """dvbs2_demod. parameters: the module's JSON keys (samplerate, symbolrate, rrc_alpha, rrc_taps, pll_bw, clock_alpha / clock_gain_omega /
    clock_mu / clock_gain_mu / clock_omega_relative_limit, modcod, shortframes, pilots, sof_thresold, ldpc_trials, agc_rate, freq_prop_factor)
    plus lut_bits / lut_phase_error: the demapper table constellation_t::make_lut(256) builds on the host (its bits [256][256][bits] int8 and
    phase errors [256][256] float32) -- data the reference computes with the host libm; hand over what the reference built (tests: the
    compiled reference's) or any table of that layout."""

    def __init__(self, parameters: dict, lut_bits: np.ndarray, lut_phase_error: np.ndarray, mem=None, capi=None, exact: bool = False, batch: int = 1):
        self.capi = capi or _capi
        self.mem = mem or TorchMem()
        p = dict(parameters)
        if "samplerate" not in p:
            raise ValueError("Samplerate parameter must be present!")          # module_demod_base.cpp:18
        if "rrc_alpha" not in p:
            raise ValueError("RRC Alpha parameter must be present!")           # module_dvbs2_demod.cpp:22
        if "pll_bw" not in p:
            raise ValueError("PLL BW parameter must be present!")              # :30
        if "modcod" not in p:
            raise ValueError("MODCOD parameter must be present!")              # :58
        if float(p.get("freq_prop_factor", 0.01)) != 0.0:
            raise NotImplementedError("dvbs2 demod mirror: freq_prop_factor must be 0 (the module's frequency feedback runs on thread timing: see the module docstring)")
        self.modcod, self.short, self.pilots = int(p["modcod"]), int(bool(p.get("shortframes", False))), int(bool(p.get("pilots", False)))
        self.loop_bw = float(p["pll_bw"])
        self.sof_thresold = float(p.get("sof_thresold", 0.6))
        self.max_trials = int(p.get("ldpc_trials", 10))
        L = self.capi.lib()
        b, s, r, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if L.sdhip_s2_cfg(self.modcod, self.short, C.byref(b), C.byref(s), C.byref(r), C.byref(c)) != 0:
            raise ValueError(self.capi.last_error())                           # get_dvbs2_cfg's messages
        self.bits, self.slots, self.rate_code = b.value, s.value, r.value
        rate = {v: k for k, v in self.capi.S2_RATES.items()}[self.rate_code]
        self.n = 16200 if self.short else 64800
        # S2PLSyncBlock's constructor (dvbs2_pl_sync.cpp:12-30)
        self.raw = (self.slots + 1) * 90
        if self.pilots:
            raw_size, cnt = self.slots - 16, 1
            while raw_size > 16:
                raw_size -= 16
                cnt += 1
            self.raw += cnt * 36
        kw = dict(samplerate=float(p["samplerate"]), symbolrate=float(p["symbolrate"]), constellation="qpsk", rrc_alpha=float(p["rrc_alpha"]),
                  rrc_taps=int(p.get("rrc_taps", 31)), agc_rate=float(p.get("agc_rate", 1e-2)), pll_bw=self.loop_bw, exact=int(exact))
        rec_alpha = 1.7e-3                                                      # module_dvbs2_demod.h:47-52
        gw, gmu = np.float32(rec_alpha ** 2 / 4.0), np.float32(rec_alpha)
        if "clock_alpha" in p:
            ca = np.float32(p["clock_alpha"])
            gw, gmu = np.float32(float(ca) ** 2 / 4.0), ca
        kw["clock_gain_omega"] = float(p.get("clock_gain_omega", gw))
        kw["clock_mu"] = float(p.get("clock_mu", 0.5))
        kw["clock_gain_mu"] = float(p.get("clock_gain_mu", gmu))
        kw["clock_omega_relative_limit"] = float(p.get("clock_omega_relative_limit", 0.005))
        for k in ("min_sps", "max_sps", "dc_block", "iq_swap", "buffer_size", "chunk_len"):
            if k in p:
                kw[k] = p[k]
        self.front = self.capi.PskDemod(self.capi.demod_cfg(**kw), front_only=True)
        self.ldpc = self.capi.LdpcDecoder(framesize=self.short, rate=rate, batch=batch)
        self.bch = self.capi.BchDecoder(framesize=self.short, rate=rate)
        self.batch = batch
        self.k = self.ldpc.info.data_len
        self.lut_bits = np.ascontiguousarray(lut_bits, dtype=np.int8)
        self.lut_phase = np.ascontiguousarray(lut_phase_error, dtype=np.float32)
        assert self.lut_bits.shape == (256, 256, self.bits) and self.lut_phase.shape == (256, 256)
        self.sym_left = np.zeros(0, dtype=np.complex64)   # the PL synchroniser's ring buffer
        self.pll_state = np.zeros(2, dtype=np.float32)
        self.soft_left = np.zeros((0, self.n), dtype=np.int8)  # frames waiting for a full decoder batch (process_s2 reads SIZE frames at a time)
        self.stats = dict(frames=0, pls=None, ldpc_trials=0.0, bch_corrections=0.0, freq=0.0)

    @property
    def bbframe_bytes(self) -> int:
        return self.bch.kbch // 8

    def process(self, iq: np.ndarray, fmt=None) -> np.ndarray:
        """One batch of baseband samples (complex64, or the integer formats with fmt) -> the BBFRAMEs it completes, uint8 [nframes, bbframe_bytes]."""
        L, m, cap = self.capi.lib(), self.mem, self.capi
        x = np.ascontiguousarray(iq)
        fmt = cap.FMT_CF32 if fmt is None else fmt
        nsamp = len(x) if x.dtype == np.complex64 else x.size // 2
        out = np.zeros((0, self.bbframe_bytes), dtype=np.uint8)
        if nsamp == 0:
            return out
        # ---- front end: clock-recovered symbols
        d_x = m.from_host(x.view(np.float32) if x.dtype == np.complex64 else x)
        d_soft = m.alloc(2 * nsamp + 64, np.int8)
        d_syms = m.alloc(2 * (nsamp + 64), np.float32)
        ns = self.front.process_dev(m.ptr(d_x), nsamp, fmt, m.ptr(d_soft), 2 * nsamp + 64, m.ptr(d_syms), nsamp + 64)
        syms = m.to_host(d_syms, 2 * (ns // 2)).view(np.complex64)
        stream = np.concatenate([self.sym_left, syms])
        if len(stream) < self.raw:
            self.sym_left = stream
            return out
        # ---- PL synchroniser
        d_st = m.from_host(stream.view(np.float32))
        cap_frames = len(stream) // self.raw + 1
        stride = self.raw
        d_fr = m.alloc(cap_frames * stride * 2, np.float32)
        consumed = C.c_size_t(0)
        nf = L.sdhip_s2_pl_sync_dev(0, self.slots, self.pilots, self.sof_thresold, C.c_void_p(m.ptr(d_st)), len(stream), C.c_void_p(m.ptr(d_fr)), stride, cap_frames,
                                    C.byref(consumed), None)
        if nf < 0:
            raise cap.SdhipError(cap.last_error())
        self.sym_left = stream[consumed.value:].copy()
        if nf == 0:
            return out
        # ---- frame PLL, soft demapper stage
        d_pl = m.alloc(cap_frames * stride * 2, np.float32)
        if L.sdhip_s2_pll_dev(0, self.modcod, self.short, self.pilots, self.loop_bw, C.c_void_p(m.ptr(d_fr)), C.c_void_p(m.ptr(d_pl)), stride, int(nf),
                              self.lut_phase.ctypes.data_as(C.c_void_p), 256, self.pll_state.ctypes.data_as(C.c_void_p)) < 0:
            raise cap.SdhipError(cap.last_error())
        d_sb = m.alloc(int(nf) * self.n, np.int8)
        d_pls = m.alloc(int(nf), np.int32)
        if L.sdhip_s2_bb_to_soft_dev(0, self.modcod, self.short, self.pilots, C.c_void_p(m.ptr(d_pl)), stride, int(nf), self.lut_bits.ctypes.data_as(C.c_void_p), 256,
                                     C.c_void_p(m.ptr(d_sb)), C.c_void_p(m.ptr(d_pls))) < 0:
            raise cap.SdhipError(cap.last_error())
        self.stats["pls"] = int(m.to_host(d_pls, int(nf))[-1])
        self.stats["freq"] = float(self.pll_state[1])
        soft = np.concatenate([self.soft_left, m.to_host(d_sb, int(nf) * self.n).reshape(int(nf), self.n)])
        # ---- LDPC in the reference's call grouping (process_s2 reads SIZE frames per decode call), repack, BCH, BB descrambler
        nfull = len(soft) // self.batch * self.batch
        self.soft_left = soft[nfull:].copy()
        if nfull == 0:
            return out
        d_w = m.from_host(soft[:nfull].reshape(-1))
        d_tr = m.alloc(nfull // self.batch, np.int32)
        self.ldpc.decode_dev(m.ptr(d_w), nfull, self.max_trials, m.ptr(d_tr))
        kb = self.k // 8
        d_pack = m.alloc(nfull * kb, np.uint8)
        d_corr = m.alloc(nfull, np.int32)
        self.bch.pack_dev(m.ptr(d_w), self.n, nfull, m.ptr(d_pack), kb)
        self.bch.decode_dev(m.ptr(d_pack), nfull, kb, m.ptr(d_corr))
        self.bch.descramble_dev(m.ptr(d_pack), nfull, kb)
        tr = m.to_host(d_tr, nfull // self.batch)
        self.stats["ldpc_trials"] = float(self.max_trials if tr[-1] == -1 else tr[-1])    # module_dvbs2_demod.cpp:254-257
        self.stats["bch_corrections"] = float(m.to_host(d_corr, nfull)[-1])
        self.stats["frames"] += nfull
        return m.to_host(d_pack, nfull * kb).reshape(nfull, kb)[:, :self.bbframe_bytes].copy()
