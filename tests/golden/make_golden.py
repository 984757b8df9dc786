#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF (oracle/_ref/libsdref.so = the
reference's own sources compiled where they lie under /root/reference, see oracle/Makefile). The reference ships no
golden vectors for this path, so these fixtures are how its behaviour is pinned where /root/reference
does not exist (the GPU box): inputs are small and stored (or regenerated from a seed and hash-checked), outputs are
what the reference produced here.

    python tests/golden/make_golden.py          # needs oracle/_ref (make -C oracle ref)

Fixtures (np.savez_compressed):
  ccdecoder.npz   CCDecoder::work: uint8 symbols -> bits, several frame sizes / noise kinds
  rs.npz          ReedSolomon::decode_interlaved on frames with 0..17 byte errors per codeword (+ the fill=-1 quirk)
  concat_*.npz    CCSDSConvConcatDecoderModule loop: int8 soft -> CADUs, per-block BER/state, RS error counts
  metop.npz       MetOpAHRPTDecoderModule loop (r=3/4 depuncture)
  demod_*.npz     PSKDemodModule chain: cs16 IQ (stored) -> int8 soft symbols + float symbols
  simple_*.npz    CCSDSSimplePSKDecoderModule loop: int8 soft -> CADUs + RS error counts (three slicer modes)
  punct_*.npz     concatenated decoder with conv_rate 3/4 and 7/8 (Viterbi_Depunc): int8 soft -> CADUs, per-block BER/state
  gardner.npz     GardnerClockRecoveryBlock on stored cs16 samples
  s2_bb_to_soft.npz  DVB-S2 S2BBToSoft (PLS decode, PL descrambling, table demapping, de-interleaver) on stored short QPSK PLFRAMEs
  ndsp_psk_*.npz  ndsp PSKDemodHierBlock (RRC -> AGC -> M&M -> Costas), stored cs16 samples -> complex symbols
  taps.npz        RRC / M&M interpolator bank / rational-resampler bank
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyref  # noqa: E402
from satdump_amd import synth  # noqa: E402
from tests import util  # noqa: E402


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def vit_symbols(rng, F, nb, kind):
    """uint8 symbol blocks for CCDecoder (127 = 0, 128 = erasure), stride 2*(F+6)."""
    stride = 2 * (F + 6)
    bits = rng.integers(0, 2, nb * F + 64).astype(np.uint8)
    coded = synth.conv_encode(bits)[: nb * 2 * F]
    if kind == "noise":
        soft = rng.integers(-127, 128, nb * 2 * F)
    elif kind == "saturated":
        soft = (coded.astype(np.int64) * 2 - 1) * 127
        soft = np.where(rng.random(len(soft)) < 0.04, -soft, soft)
    else:
        amp, sig = {"clean": (60, 20), "noisy": (50, 45)}[kind]
        soft = np.clip(np.rint((coded.astype(float) * 2 - 1) * amp + rng.standard_normal(len(coded)) * sig), -127, 127).astype(np.int64)
    u = soft + 127
    u[u == 128] = 127
    if kind == "noisy":
        u[rng.random(len(u)) < 0.1] = 128
    syms = np.full(nb * stride, 128, dtype=np.uint8)
    for b in range(nb):
        syms[b * stride: b * stride + 2 * F] = u[b * 2 * F:(b + 1) * 2 * F]
    return syms


def main():
    ref = pyref.ref()
    out = {}

    # ---- CCDecoder
    d = {}
    for i, (F, nb, kind) in enumerate([(4096, 3, "clean"), (4096, 3, "noisy"), (1024, 4, "noise"), (640, 5, "saturated"), (12288, 2, "noisy")]):
        rng = np.random.default_rng(100 + i)
        syms = vit_symbols(rng, F, nb, kind)
        bits = ref.ccdecoder(F, syms)
        d[f"c{i}_F"] = np.int32(F)
        d[f"c{i}_syms"] = syms
        d[f"c{i}_bits"] = np.packbits(bits)
    out["ccdecoder"] = d

    # ---- RS
    rng = np.random.default_rng(7)
    frames = synth.make_cadus(24, seed=7, derand=False)
    bad = frames.copy()
    for f in range(len(bad)):
        for cw in range(4):
            ne = (f + cw * 5) % 19  # 0..18 errors: beyond t=16 -> uncorrectable (-1)
            pos = rng.choice(255, ne, replace=False)
            for p in pos:
                bad[f, 4 + p * 4 + cw] ^= rng.integers(1, 256)
    dec, err = ref.rs_decode(bad, I=4, dualbasis=True, fill_bytes=-1)
    dec0, err0 = ref.rs_decode(bad, I=4, dualbasis=True, fill_bytes=0)
    out["rs"] = dict(frames=bad, dec=dec, err=err, dec_fill0=dec0, err_fill0=err0)

    # ---- concatenated decoder, BPSK (GOES) and QPSK (NPP), incl. a polarity inversion + garbage gap (sync loss)
    for name, const, oconst, sigma in [("bpsk", "bpsk", pyref.BPSK, 28.0), ("qpsk", "qpsk", pyref.QPSK, 55.0)]:
        spec = synth.SynthSpec(constellation=const, samplerate=3e6, symbolrate=1e6, nrzm=True, seed=21)
        cadus = synth.make_cadus(14, seed=21)
        syms = synth.frames_to_symbols(cadus, spec)
        soft = synth.soft_from_symbols(syms, spec, sigma=sigma, seed=21)
        rng = np.random.default_rng(5)
        gap = rng.integers(-127, 128, 3 * 8192).astype(np.int8)
        half = (len(soft) // 2) // 8192 * 8192
        soft = np.concatenate([soft[:half], gap, (-soft[half:].astype(np.int16)).clip(-127, 127).astype(np.int8)])
        cfg = pyref.fec_cfg(constellation=oconst, nrzm=1, rs_usecheck=1)
        r = ref.concat_decode(cfg, soft)
        out[f"concat_{name}"] = dict(soft=soft, cadu=r["cadu"], ber=r["ber"], state=r["state"], frm_err=r["frm_err"])

    # ---- MetOp r=3/4
    spec, cadus, plain, syms = util.metop_case(nframes=12)
    soft = synth.soft_from_symbols(syms, spec, sigma=40.0, seed=3)
    r = ref.metop_decode(soft)
    out["metop"] = dict(soft=soft, cadu=r["cadu"], ber=r["ber"], state=r["state"], frm_err=r["frm_err"])

    # ---- psk_demod chain on cs16 input (the stored samples ARE the input: nothing depends on numpy's RNG)
    for name, mk, ocfg, n in [
        ("goes", lambda: util.goes_case(nframes=5), pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0), 120000),
        ("metop", lambda: util.metop_case(nframes=8), pyref.demod_cfg(), 120000),
        ("npp", lambda: util.npp_case(nframes=8), pyref.demod_cfg(samplerate=30e6, symbolrate=15e6, constellation=pyref.QPSK, pll_bw=0.002), 120000),
    ]:
        spec, cadus, plain, syms = mk()
        x, _ = synth.modulate(syms, spec)
        cs16 = synth.to_cs16(x[:n])
        xin = (cs16.astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)  # baseband_interface.h:178 (volk_16i_s32f_convert_32f)
        r = ref.psk_demod(ocfg, xin)
        out[f"demod_{name}"] = dict(cs16=cs16, soft=r["soft"], syms=r["syms"], buffer_size=np.int32(r["buffer_size"]), final_sps=np.float32(r["final_sps"]))

    # ---- ccsds_simple_psk_decoder: BPSK + NRZ-M, QPSK differential, QPSK two-deframer case (90 degree stream)
    for name in ("bpsk_nrzm", "qpsk_diff_swap", "qpsk_90deg"):
        ck, soft, plain = util.simple_case(name, sigma=18.0, nframes=8, seed=31)
        ock = dict(ck)
        ock["constellation"] = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK}[ck["constellation"]]
        r = ref.simple_decode(pyref.fec_cfg(decoder=2, rs_usecheck=0, **ock), soft)
        out[f"simple_{name}"] = dict(soft=soft, cadu=r["cadu"], frm_err=r["frm_err"])

    # ---- conv_rate != 1/2: Viterbi_Depunc behind the concatenated decoder's loop (SURVEY.md 8 row a13'), 3/4 and 7/8, with a noise gap
    for rate, name, sig in [(2, "r34", 20.0), (4, "r78", 11.0)]:
        soft, plain = util.punctured_case(rate, nframes=8, sigma=sig, seed=40 + rate, gap=(rate == 2))
        r = ref.concat_decode_punc(pyref.fec_cfg(constellation=pyref.QPSK, nrzm=0, rs_usecheck=1), rate, soft)
        out[f"punct_{name}"] = dict(soft=soft, rate=np.int32(rate), cadu=r["cadu"], ber=r["ber"], state=r["state"], frm_err=r["frm_err"])

    # ---- Gardner clock recovery block (clock_recovery_gardner.cpp) on a QPSK baseband
    spec, cadus, plain, syms = util.metop_case(nframes=3)
    x, _ = synth.modulate(syms, spec)
    cs16 = synth.to_cs16(x[:40000])
    xin = (cs16.astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)
    gp = np.array([2.5714, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005], dtype=np.float32)
    out["gardner"] = dict(cs16=cs16, params=gp, syms=ref.block(7, gp, xin))

    # ---- psk_demod's has_carrier chain (carrier PLL + DC block in front of the Costas loop) and the carrier PLL block alone
    from tests.test_zy_demod_additions_gpu import _carrier_case
    x, kw = _carrier_case(nframes=6)
    cs16 = synth.to_cs16(x[:120000])
    xin = (cs16.astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)
    r = ref.psk_demod(pyref.demod_cfg(constellation=pyref.BPSK, **kw), xin)
    pp = np.array([0.002, 3.14, -3.14], dtype=np.float32)
    out["demod_carrier"] = dict(cs16=cs16, soft=r["soft"], syms=r["syms"], pll_params=pp, pll_out=ref.block(8, pp, xin[:40000]))

    # ---- ndsp PSK demodulator hier block (src-core/dsp/hier/psk_demod.h), the reference's own threads and FIFOs: stored cs16 -> symbols
    from tests.test_ndsp_gpu import _signal
    nd = pyref.NdspRef()
    for const, sr, symr in (("qpsk", 6e6, 2.33e6), ("bpsk", 6e6, 2e6)):
        cs16 = synth.to_cs16(_signal(const, 12000, sr, symr)[:30000])
        xin = (cs16.astype(np.float32) * np.float32(1.0 / 32767.0)).view(np.complex64)
        out[f"ndsp_psk_{const}"] = dict(cs16=cs16, samplerate=np.float64(sr), symbolrate=np.float64(symr),
                                        syms=nd.run("psk_demod_cc", {"constellation": const, "samplerate": sr, "symbolrate": symr}, xin))

    # ---- DVB-S2 PLFRAME -> soft bits stage (dvbs2::S2BBToSoft driven through its own streams, oracle/ref_wrap_dvbs2_demap.cpp): short QPSK
    # frames (MODCOD 4), stored samples + the demapper table the reference builds -> PLS index and the LDPC decoder's input
    from tests import dvbs2_util
    fr_ref = pyref.S2FrontRef()
    c4 = fr_ref.cfg(4, 1, 0)
    frames = dvbs2_util.plframes(c4["slots"], (4 << 2) | 2, 2, seed=11, stride_pad=6)
    soft4, pls4 = fr_ref.bb_to_soft(4, 1, 0, frames)
    out["s2_bb_to_soft"] = dict(frames=frames, modcod=np.int32(4), shortframes=np.int32(1), pilots=np.int32(0), lut=fr_ref.lut(4, 1), soft=soft4, pls=pls4)

    # ---- filter designs
    bank, ir, dr = ref.resamp_bank(2700000, 3000000)
    out["taps"] = dict(rrc_goes=ref.rrc_taps(2.7e6, 927000, 0.5, 31), rrc_metop=ref.rrc_taps(6e6, 2333333, 0.5, 31), mm=ref.mm_bank(128, 8),
                       resamp=bank, resamp_ratio=np.array([ir, dr], dtype=np.int32))

    # `make_golden.py name ...` rewrites only the named fixtures (and their INDEX lines): an .npz is a zip with time stamps, so
    # rewriting an unchanged fixture would still change its bytes
    only = set(sys.argv[1:])
    old_lines = {}
    ipath = os.path.join(HERE, "INDEX.txt")
    if only and os.path.exists(ipath):
        for ln in open(ipath).read().splitlines():
            if ln and not ln.startswith("#"):
                old_lines[ln.split(".npz")[0]] = ln
    index = []
    for name, d in out.items():
        path = os.path.join(HERE, name + ".npz")
        if only and name not in only:
            if name in old_lines:
                index.append(old_lines[name])
            continue
        np.savez_compressed(path, **d)
        index.append(f"{name}.npz  {os.path.getsize(path):8d} B  " + " ".join(f"{k}:{sha(v)[:12]}" for k, v in sorted(d.items())))
    with open(ipath, "w") as f:
        f.write("# produced by tests/golden/make_golden.py from oracle/_ref (the reference's own code); key:sha256[:12]\n" + "\n".join(index) + "\n")
    print("\n".join(index))


if __name__ == "__main__":
    main()
