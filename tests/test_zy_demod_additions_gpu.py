"""Demodulator GPU tests added after the last GPU visit of round 1 (validated on the host twin only so far, tests/test_demod_gpu_on_twin_cpu.py
collects them too). The file sorts behind the long-standing GPU tests on purpose."""
import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth
from tests.test_demod_gpu import _run_demod, capi, orc, torch_cuda  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def test_dc_block_in_front_of_the_resampler(torch_cuda, capi, orc):
    """dc_block=1 together with the rational resampler (samplerate / symbolrate above max_sps): the resampler must read the DC-blocked
    samples. (It read the stage's input: found by the differential fuzz on the host twin, tests/test_demod_emu_cpu.py.)"""
    rng = np.random.default_rng(77)
    n = 60000
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3 + (0.05 - 0.02j)).astype(np.complex64)
    kw = dict(samplerate=19153000.0, symbolrate=3.5e6, rrc_alpha=0.5, rrc_taps=21, pll_bw=0.006, dc_block=1)
    want = orc.psk_demod(pyref.demod_cfg(constellation=pyref.QPSK, **kw), x)
    soft, syms, st = _run_demod(torch_cuda, capi, dict(constellation="qpsk", **kw), x, chunks=[0, 12345, n], exact=1)
    assert st.final_sps == np.float32(want["final_sps"]) and want["final_sps"] < 4.0
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32))
    assert np.array_equal(soft, want["soft"])


def test_dc_block_chunk_parallel(torch_cuda, capi, orc):
    """dc_block=1 in the default chunk-parallel mode: the accumulator at every chunk start comes from an affine scan in double,
    every chunk then runs the reference's float recurrence (correct_iq.cpp:27-31) and is certified against its predecessor
    within 1e-5 |acc|. A MetOp stream with a DC offset of a quarter of its amplitude, several calls: same symbol count, >= 99 %
    of the float symbols within 1e-5 of the sequential reference's, CADUs identical -- and the stage really ran in chunks."""
    from tests import test_demod_gpu as G
    spec, plain, x, ocfg, kw, fec, ofec = G._case("metop")
    x = (x + np.complex64(0.06 - 0.03j)).astype(np.complex64)
    ocfg.dc_block = 1
    want = orc.psk_demod(ocfg, x)
    n = len(x)
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, dc_block=1), x, chunks=[0, n // 2 + 77, n], chunk_len=8192)
    assert st.chunks >= 4 * ((n - n // 2 - 77) // 8192)  # last call: dc + agc + costas + mm stages, all chunked
    assert len(syms) == len(want["syms"])
    ref = want["syms"]
    err = np.abs(syms - ref) / np.sqrt(np.mean(np.abs(ref) ** 2))
    assert np.mean(err > 1e-5) < 0.01 and np.median(err) < 2e-6, (float(np.mean(err > 1e-5)), float(np.median(err)))
    dec = capi.FecDecoder(capi.fec_cfg(**fec))
    dec.push(soft)
    got = dec.pull()
    wantc = orc.metop_decode(want["soft"])["cadu"]
    assert got.shape == wantc.shape and np.array_equal(got, wantc) and len(got) >= 20


@pytest.mark.parametrize("samplerate,symbolrate,max_sps,what", [
    (6e6, 927000, 3.0, "decimate by 2, then 9/10"),            # 6 Msps recording of the GOES HRIT signal
    (12e6, 1000000, 3.0, "decimate by 4, nothing else"),       # exact power of two: plan_4 = two half-band stages
    (33e6, 927000, 3.0, "decimate by 8, then 54/81-ish"),      # wideband recording: plan_8 (4 + 2)
])
def test_power_of_two_predecimator(torch_cuda, capi, orc, samplerate, symbolrate, max_sps, what):
    """SmartResampler's power-of-two pre-decimator (smart_resampler.cpp:15-29, power_decim.cpp, the plans' tap tables) in front of
    the rational resampler: exact mode bit-identical to the reference over several ragged calls (stage phases and histories
    carry), and the chunk-parallel mode delivers the same symbol count."""
    from satdump_amd import synth
    spec = synth.SynthSpec(constellation="bpsk", samplerate=samplerate, symbolrate=symbolrate, conv="1/2", nrzm=True, esn0_db=9.0, amplitude=0.4,
                           cfo_hz=2000.0, seed=11)
    cadus = synth.make_cadus(6, seed=11)
    x, _ = synth.modulate(synth.frames_to_symbols(cadus, spec), spec)
    x = x[:600000]
    kw = dict(samplerate=samplerate, symbolrate=symbolrate, rrc_alpha=0.5, pll_bw=0.02, max_sps=max_sps)
    want = orc.psk_demod(pyref.demod_cfg(constellation=pyref.BPSK, **kw), x)
    n = len(x)
    soft, syms, st = _run_demod(torch_cuda, capi, dict(constellation="bpsk", **kw), x, chunks=[0, 1, 4097, 100001, n], exact=1)
    assert st.final_sps == np.float32(want["final_sps"]) and len(want["syms"]) > 10000
    assert len(syms) == len(want["syms"])
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32)), what
    assert np.array_equal(soft, want["soft"])
    soft2, syms2, st2 = _run_demod(torch_cuda, capi, dict(constellation="bpsk", **kw), x, chunk_len=4096)
    assert len(soft2) == len(want["soft"]) and np.mean(soft2 != want["soft"]) < 0.02


@pytest.mark.parametrize("samplerate,symbolrate,custom,max_sps,what", [
    (3e6, 927000, 2500000, 3.0, "input sps 3.24 outside [1.1, 3]: the resampler runs, at 5/6 instead of initb's 9/10"),
    (3e6, 1200000, 2880000, 3.0, "input sps 2.5 inside the range: NO resampler, the chain's rates (RRC taps, omega, offset limit) are the custom one's all the same"),
])
def test_custom_samplerate(torch_cuda, capi, orc, samplerate, symbolrate, custom, max_sps, what):
    """"custom_samplerate" (module_demod_base.cpp:73-74; VERDICT r5 missing 4): final_samplerate is the parameter's value, the resample decision stays initb's.
    Exact mode bit-identical to the reference's chain (compiled in place) over ragged calls, the chunk-parallel mode delivers the same symbol count."""
    from satdump_amd import synth
    spec = synth.SynthSpec(constellation="bpsk", samplerate=samplerate, symbolrate=symbolrate, conv="1/2", nrzm=True, esn0_db=9.0, amplitude=0.4, cfo_hz=1500.0, seed=12)
    x, _ = synth.modulate(synth.frames_to_symbols(synth.make_cadus(6, seed=12), spec), spec)
    x = x[:400000]
    kw = dict(samplerate=samplerate, symbolrate=symbolrate, rrc_alpha=0.5, pll_bw=0.02, max_sps=max_sps, custom_samplerate=custom)
    want = orc.psk_demod(pyref.demod_cfg(constellation=pyref.BPSK, **kw), x)
    n = len(x)
    soft, syms, st = _run_demod(torch_cuda, capi, dict(constellation="bpsk", **kw), x, chunks=[0, 1, 4097, 100001, n], exact=1)
    assert st.final_samplerate == np.float32(custom) and st.final_sps == np.float32(want["final_sps"]) and len(want["syms"]) > 10000, what
    assert len(syms) == len(want["syms"])
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32)), what
    assert np.array_equal(soft, want["soft"])
    if st.resample_interp:  # (without the resampler the custom rate is NOT the signal's: the timing loop sits on its rate limit, unlocked -- no time-parallel schedule
        #                      follows that symbol for symbol; the exact mode above is what pins the parameter there)
        soft2, syms2, st2 = _run_demod(torch_cuda, capi, dict(constellation="bpsk", **kw), x, chunk_len=4096)
        assert len(soft2) == len(want["soft"]) and np.mean(soft2 != want["soft"]) < 0.02


@pytest.mark.parametrize("case", ["goes", "npp"])
def test_post_costas_dc(torch_cuda, capi, orc, case):
    """psk_demod's post_costas_dc option (module_psk_demod.cpp:36-38, 127-134; NOAA / Psyche / Stereo pipelines): a DC block between
    the Costas loop and the clock recovery. Exact mode bit-identical over ragged calls; chunk-parallel mode (Costas chunks turned
    back into one frame in front of the DC block) same symbol count, >= 98.5 % of the symbols within 1e-5, CADUs identical."""
    from tests import test_demod_gpu as G
    spec, plain, x, ocfg, kw, fec, ofec = G._case(case)
    ocfg.post_costas_dc = 1
    want = orc.psk_demod(ocfg, x)
    n = len(x)
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, post_costas_dc=1), x[:300000], chunks=[0, 1000, 77777, 300000], exact=1)
    w3 = orc.psk_demod(ocfg, x[:300000])
    assert np.array_equal(syms.view(np.uint32), w3["syms"].view(np.uint32)) and np.array_equal(soft, w3["soft"])
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, post_costas_dc=1), x, chunks=[0, n // 2 + 3, n], chunk_len=8192)
    assert len(syms) == len(want["syms"])
    err = np.abs(syms - want["syms"]) / np.sqrt(np.mean(np.abs(want["syms"]) ** 2))
    assert np.mean(err > 1e-5) < 0.015, float(np.mean(err > 1e-5))
    dec = capi.FecDecoder(capi.fec_cfg(**fec))
    dec.push(soft)
    got = dec.pull()
    wantc = orc.concat_decode(ofec, want["soft"])["cadu"]
    assert got.shape == wantc.shape and np.array_equal(got, wantc) and len(got) >= 20


@pytest.mark.parametrize("q8", ["0", "1"])
@pytest.mark.parametrize("case", ["goes", "metop"])
def test_soft_symbols_without_the_float_symbols(torch_cuda, capi, orc, case, q8, monkeypatch):
    """A caller that does not ask for the float symbols (the plugin, bench.py's timed steps) gets its .soft bytes from the
    quantiser's vectorised path (four symbols, one aligned store) -- and, with SDHIP_MM_Q8=1 (an experiment that stays off by
    default, see DemodEngine), from the clock-recovery kernel itself, two bytes per symbol through the per-chunk scratch. Either way
    they must be the very bytes the path with float symbols produces, over several ragged calls (rows start at every output
    alignment), in exact and in chunk-parallel mode."""
    from tests import test_demod_gpu as G
    monkeypatch.setenv("SDHIP_MM_Q8", q8)
    spec, plain, x, ocfg, kw, fec, ofec = G._case(case)
    x = x[:700000]
    n = len(x)
    d_x = G._dev(torch_cuda, x.view(np.float32))
    for extra in (dict(exact=1), dict(chunk_len=8192)):
        bounds = [0, 1000, n // 3 + 5, n]
        soft_f, _, _ = _run_demod(torch_cuda, capi, kw, x, chunks=bounds, **extra)
        dem = capi.PskDemod(capi.demod_cfg(**kw, **extra))
        parts = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            d_soft = torch_cuda.zeros(2 * (b - a) + 64, dtype=torch_cuda.int8, device="cuda")
            ns = dem.process_dev(d_x.data_ptr() + 8 * a, b - a, capi.FMT_CF32, d_soft.data_ptr(), 2 * (b - a) + 64)  # no float symbols wanted
            parts.append(d_soft[:ns].cpu().numpy())
        soft_q = np.concatenate(parts)
        assert len(soft_q) == len(soft_f) > 100000 and np.array_equal(soft_q, soft_f), extra


def _carrier_case(nframes=24, carrier=0.9, cfo_hz=3000.0):
    """BPSK with a residual carrier in quadrature (an AM-subcarrier style downlink, the has_carrier pipelines): the GOES test stream
    built without frequency offset, the carrier line added, then both turned by the offset."""
    import dataclasses
    from tests import util
    spec, cadus, plain, syms = util.goes_case(nframes=nframes)
    spec0 = dataclasses.replace(spec, cfo_hz=0.0, phase0=0.0)
    x0, _ = synth.modulate(syms, spec0)
    t = np.arange(len(x0), dtype=np.float64)
    rot = np.exp(1j * (2 * np.pi * cfo_hz / spec.samplerate * t + 0.4))
    x = ((x0.astype(np.complex128) + 1j * carrier * spec.amplitude) * rot).astype(np.complex64)
    kw = dict(samplerate=3e6, symbolrate=927000, rrc_alpha=0.5, pll_bw=0.006, max_sps=3.0, has_carrier=1, carrier_pll_bw=0.002)
    return x, kw


def test_has_carrier(torch_cuda, capi, orc):
    """psk_demod's has_carrier mode (module_psk_demod.cpp:39-40, 93-113; the ODIN pipeline): carrier-tracking PLL
    (pll_carrier_tracking.cpp, the table-driven arctangent and the polynomial sine / cosine of fast_trig.cpp) and a DC block between
    the RRC filter and the Costas loop, whose frequency limit drops to 0.2. Exact mode bit-identical to the reference over ragged
    calls; chunk-parallel mode (PLL and DC block as speculative chunk stages) same symbol count, >= 98.5 % of the float symbols
    within 1e-5, CADUs identical; a non-BPSK constellation is refused like the reference refuses it."""
    x, kw = _carrier_case()
    ocfg = pyref.demod_cfg(constellation=pyref.BPSK, **kw)
    want = orc.psk_demod(ocfg, x)
    assert len(want["syms"]) > 100000
    ofec = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=1)
    wantc = orc.concat_decode(ofec, want["soft"])["cadu"]
    assert len(wantc) >= 18  # the reference itself decodes the stream: the carrier loop locked
    w3 = orc.psk_demod(ocfg, x[:300000])
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, constellation="bpsk"), x[:300000], chunks=[0, 1000, 77777, 300000], exact=1)
    assert np.array_equal(syms.view(np.uint32), w3["syms"].view(np.uint32)) and np.array_equal(soft, w3["soft"])
    n = len(x)
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, constellation="bpsk"), x, chunks=[0, n // 2 + 3, n], chunk_len=8192)
    assert st.chunks >= 5 * (int(0.9 * (n - n // 2 - 3)) // 8192 - 1)  # last call, 9/10-resampled: agc, pll, dc, costas, mm all chunked
    assert len(syms) == len(want["syms"])
    err = np.abs(syms - want["syms"]) / np.sqrt(np.mean(np.abs(want["syms"]) ** 2))
    assert np.mean(err > 1e-5) < 0.015, float(np.mean(err > 1e-5))
    dec = capi.FecDecoder(capi.fec_cfg(constellation="bpsk", nrzm=1, rs_i=4, rs_type=1, rs_usecheck=1))
    dec.push(soft)
    got = dec.pull()
    assert got.shape == wantc.shape and np.array_equal(got, wantc)
    with pytest.raises(capi.SdhipError):
        capi.PskDemod(capi.demod_cfg(**dict(kw, constellation="qpsk")))
    with pytest.raises(capi.SdhipError):
        capi.PskDemod(capi.demod_cfg(**dict(kw, constellation="bpsk", carrier_pll_bw=0.0)))


@pytest.mark.parametrize("workload,frames", [("metop_ahrpt", 2100), ("npp_hrd", 2000)])
def test_three_passes_through_stateful_handles_equal_the_reference_on_the_tiled_recording(torch_cuda, capi, orc, workload, frames):
    """What bench.py times at N=1: ONE pair of handles, the periodic recording passed through it step after step (engine state,
    histories, pending FEC buffers, Costas frame all carry over the call boundary). Three passes of a 29-33 M-sample recording
    against the reference run once over the recording tiled three times: the CADU lists must be identical byte for byte -- frames
    around every call boundary included, and, for MetOp (no RS check), the frames RS cannot correct as well."""
    import bench
    wl = bench.WORKLOADS[workload]
    rec = synth.Recording(synth.SynthSpec(**wl["spec"]), frames, blocks=1)
    x = rec.synth_range(0, rec.n_samples, device="cuda")
    n = x.numel()
    dem = capi.PskDemod(capi.demod_cfg(**wl["demod"]))
    fec = capi.FecDecoder(capi.fec_cfg(**wl["fec"]))
    d_soft = torch_cuda.empty(2 * n + 64, dtype=torch_cuda.int8, device="cuda")
    d_cadu = torch_cuda.empty((frames + 64, 1024), dtype=torch_cuda.uint8, device="cuda")
    outs, nsoft = [], 0
    for _ in range(3):
        ns = dem.process_dev(x.data_ptr(), n, capi.FMT_CF32, d_soft.data_ptr(), 2 * n + 64)
        nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), frames + 64)
        outs.append(d_cadu[:nf].cpu().numpy().copy())
        nsoft += ns
    got = np.concatenate(outs)
    xh = x.cpu().numpy()
    r, refc, _, _ = bench.ref_decode(orc, wl, np.concatenate([xh, xh, xh]), want_syms=False)
    assert nsoft == len(r["soft"])
    assert got.shape == refc.shape and np.array_equal(got, refc)
    assert len(outs[1]) >= frames - 1 and len(outs[2]) >= frames - 1


@pytest.mark.parametrize("case,shift", [("metop", 250000), ("goes", -37500)])
def test_freq_shift(torch_cuda, capi, ref, case, shift):
    """`freq_shift` (dsp::FreqShiftBlock between the DC block and the resampler, module_demod_base.cpp:122-123; VOLK's rotator2, restated in
    ref_shim/volk/volk.h -- VOLK is not part of the reference tree -- incl. its renormalisation every 512 samples and at the end of every
    source buffer): a recording that sits `shift` Hz off. exact = 1: the sequential recurrence, soft AND float symbols bit-identical over
    ragged calls (the rotator's position in its source buffer carries across them). Chunk-parallel mode: the closed form (exact phase in
    64-bit fixed-point turns, the renormalisation's amplitude sawtooth kept); what the reference's float recurrence adds is a slowly
    varying phase the carrier loop tracks out: same symbol count, the usual >= 99 % within 1e-5, CADUs identical."""
    from tests import test_demod_gpu as G
    spec, plain, x, ocfg, kw, fec, ofec = G._case(case)
    n = len(x)
    x = (x * np.exp(-2j * np.pi * shift / kw["samplerate"] * np.arange(n))).astype(np.complex64)
    ocfg.freq_shift = float(shift)
    want = ref.psk_demod(ocfg, x)
    wantc = (ref.metop_decode(want["soft"]) if case == "metop" else ref.concat_decode(ofec, want["soft"]))["cadu"]
    assert len(wantc) >= 20
    cuts = [0, 5, 77777, n // 2 + 1, n]
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, freq_shift=float(shift)), x, chunks=cuts, exact=1)
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32)) and np.array_equal(soft, want["soft"])
    soft, syms, st = _run_demod(torch_cuda, capi, dict(kw, freq_shift=float(shift)), x, chunks=[0, n // 3, n], chunk_len=8192)
    assert len(syms) == len(want["syms"]) and st.chunks > 30
    err = np.abs(syms - want["syms"]) / np.sqrt(np.mean(np.abs(want["syms"]) ** 2))
    assert np.mean(err > 1e-5) < (0.012 if case == "goes" else 0.007), float(np.mean(err > 1e-5))
    dec = capi.FecDecoder(capi.fec_cfg(**fec))
    dec.push(soft)
    got = dec.pull()
    assert got.shape == wantc.shape and np.array_equal(got, wantc)


def test_dvbs2_front_end(torch_cuda, capi, orc):
    """The DVB-S2 demodulator's front end (module_dvbs2_demod.cpp:98-105: AGC, RRC filter, M&M clock recovery -- psk_demod's stages without a
    Costas loop, sdhip_dvbs2_front_create): exact mode bit-identical to the reference's three blocks chained (each compiled in place),
    over ragged calls; the chunk-parallel mode delivers the same symbol count within the clock recovery's floor."""
    import ctypes as C
    from tests import dvbs2_util
    raw = 91 * 90
    cw = np.random.default_rng(2).integers(0, 2, (12, 16200)).astype(np.uint8)
    sym = dvbs2_util.pl_stream_from_bits(cw, raw, (4 << 2) | 2, seed=8, lead=0, cfo=0.002, esn0_db=11.0)  # (signal from the first sample: in noise the timing loop has no trajectory two schedules could share)
    # pulse-shape the symbol stream to 3 samples per symbol (rolloff 0.35)
    h = synth.rrc_impulse(3.0, 0.35, 10) * np.sqrt(3.0)
    up = np.zeros(len(sym) * 3, dtype=np.complex128)
    up[::3] = sym
    x = (np.convolve(up, h, mode="same") * 0.5).astype(np.complex64)
    agc, rrc, mm = [1e-3, 1.0, 1.0, 65536.0], [3e6, 1e6, 0.35, 31], [3.0, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005]
    want = orc.block(3, mm, orc.block(1, rrc, orc.block(0, agc, x)))
    kw = dict(samplerate=3e6, symbolrate=1e6, constellation="qpsk", rrc_alpha=0.35, rrc_taps=31, agc_rate=1e-3, pll_bw=0.005)
    n = len(x)
    d_x = torch_cuda.from_numpy(np.ascontiguousarray(x.view(np.float32))).cuda()

    def run(bounds, **extra):
        dem = capi.PskDemod(capi.demod_cfg(**kw, **extra), front_only=True)
        out = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            m = b - a
            d_soft = torch_cuda.zeros(2 * m + 64, dtype=torch_cuda.int8, device="cuda")
            d_syms = torch_cuda.zeros(2 * (m + 64), dtype=torch_cuda.float32, device="cuda")
            ns = dem.process_dev(d_x.data_ptr() + 8 * a, m, capi.FMT_CF32, d_soft.data_ptr(), 2 * m + 64, d_syms.data_ptr(), m + 64)
            out.append(d_syms[: 2 * (ns // 2)].cpu().numpy().view(np.complex64))
        return np.concatenate(out), dem.stats()

    got, st = run([0, 1000, 77777, n], exact=1)
    assert len(got) == len(want) and len(want) > 100000 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    got, st = run([0, n // 2 + 3, n], chunk_len=8192)
    assert st.chunks > 20 and len(got) == len(want)
    err = np.abs(got - want) / np.sqrt(np.mean(np.abs(want) ** 2))
    assert np.mean(err > 1e-5) < 0.02 and np.median(err) < 2e-6, (float(np.mean(err > 1e-5)), float(np.median(err)))


@pytest.mark.parametrize("case", ["metop", "goes"])
def test_doppler(torch_cuda, capi, orc, case):
    """`enable_doppler` (dsp::DopplerCorrectBlock behind the frequency shift, module_demod_base.cpp:125-171): a recording whose carrier sweeps through
    +-8 kHz like a pass does. The block's target frequency comes from the pass prediction once per source buffer (SGP4 on a TLE: host work that stays with
    the caller, sdhip_demod_doppler_targets); the rotator that ramps toward it is the device's. exact = 1: the block's sample loop (oracle/sd_oracle.c's restatement of
    doppler_correct.cpp:41-63, pinned to the compiled block by tests/test_oracle_vs_ref.py) in front of the reference chain: soft AND
    float symbols bit-identical over ragged calls (rotator state, buffer position and the queue of targets carry across them). Chunk-parallel mode: the
    recurrence in closed form in double. What that leaves out is the rounding NOISE of the reference's float phase accumulation (phase += freq on a float
    of magnitude up to 2 pi: +-2.4e-7 rad per sample, a random walk of ~2e-6 rad over the carrier loop's memory, which the loop cannot track out): same
    symbol count, median error 3e-6 (measured; 3.5e-7 without a sweep), 13 % of the symbols between 1e-5 and 1e-4, ~1 % beyond 1e-4 (GOES at 7 dB: the clock recovery's own floor), CADUs identical.
    No time-parallel schedule can follow a float accumulator's rounding sequence (the recurrence is additive: two trajectories never contract onto each
    other); exact = 1 does. A call for which no target is queued fails loudly."""
    from oracle import pyref
    from tests import test_demod_gpu as G
    spec, plain, x, ocfg, kw, fec, ofec = G._case(case)
    n = len(x)
    fs = kw["samplerate"]
    t = np.arange(n) / fs
    f_dop = 8000.0 * np.cos(np.pi * np.arange(n) / n)  # +8 kHz -> -8 kHz over the recording
    x = (x * np.exp(2j * np.pi * np.cumsum(f_dop) / fs)).astype(np.complex64)
    st0 = capi.PskDemod(capi.demod_cfg(**kw)).stats()
    buf = st0.buffer_size
    nbuf = n // buf + 2
    targets = np.array([-2 * np.pi * f_dop[min(n - 1, (k + 1) * buf - 1)] / fs for k in range(nbuf)], dtype=np.float32)  # hz_to_rad(-doppler_shift, samplerate) behind buffer k
    y, _ = pyref.doppler_ref(x, 0.01, buf, targets)
    want = orc.psk_demod(ocfg, y)
    wantc = (orc.metop_decode(want["soft"]) if case == "metop" else orc.concat_decode(ofec, want["soft"]))["cadu"]
    assert len(wantc) >= 20

    def run(cuts, **extra):
        dem = capi.PskDemod(capi.demod_cfg(**dict(kw, doppler=1, doppler_alpha=0.01), **extra))
        dem.doppler_targets(targets[:3])
        dem.doppler_targets(targets[3:])
        d_x = G._dev(torch_cuda, x.view(np.float32))
        soft, syms = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            m = b - a
            d_soft = torch_cuda.zeros(2 * m + 64, dtype=torch_cuda.int8, device="cuda")
            d_syms = torch_cuda.zeros(2 * (m + 64), dtype=torch_cuda.float32, device="cuda")
            ns = dem.process_dev(d_x.data_ptr() + 8 * a, m, capi.FMT_CF32, d_soft.data_ptr(), 2 * m + 64, d_syms.data_ptr(), m + 64)
            nsym = ns if kw["constellation"] == "bpsk" else ns // 2
            soft.append(d_soft[:ns].cpu().numpy())
            syms.append(d_syms[: 2 * nsym].cpu().numpy().view(np.complex64))
        return np.concatenate(soft), np.concatenate(syms), dem.stats()

    soft, syms, st = run(sorted({c for c in (0, 5, buf, buf + 1, 3 * buf, 77777, n // 2 + 1) if c < n} | {n}), exact=1)
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32)) and np.array_equal(soft, want["soft"])
    soft, syms, st = run([0, n // 3, n], chunk_len=8192)
    assert len(syms) == len(want["syms"]) and st.chunks > 30
    err = np.abs(syms - want["syms"])[len(syms) // 10:] / np.sqrt(np.mean(np.abs(want["syms"]) ** 2))
    assert np.median(err) < 1e-5 and np.mean(err > 1e-4) < (0.02 if case == "goes" else 0.012), (float(np.median(err)), float(np.mean(err > 1e-4)))
    dec = capi.FecDecoder(capi.fec_cfg(**fec))
    dec.push(soft)
    got = dec.pull()
    assert got.shape == wantc.shape and np.array_equal(got, wantc)
    dem = capi.PskDemod(capi.demod_cfg(**dict(kw, doppler=1)))
    d_x = G._dev(torch_cuda, x.view(np.float32))
    d_soft = torch_cuda.zeros(2 * n + 64, dtype=torch_cuda.int8, device="cuda")
    with pytest.raises(capi.SdhipError, match="targets"):
        dem.process_dev(d_x.data_ptr(), n, capi.FMT_CF32, d_soft.data_ptr(), 2 * n + 64)


@pytest.mark.parametrize("case,chunk", [("goes", 8192), ("metop", 16384), ("npp", 8192)])
def test_cooperative_lanes_equal_the_per_lane_streams(torch_cuda, capi, orc, case, chunk, monkeypatch):
    """Round 5: the lane stages move their 64 streams cooperatively (demod_kernels.hip, Coop: eight whole 128-byte lines per load / store instruction, handed to
    their owners through an LDS transpose) instead of 64 sixteen-byte pieces of 64 lines. Same bytes, same arithmetic: soft symbols, float symbols and chunk
    statistics must be BIT-identical to the per-lane path (SDHIP_COOP=0), on streams long enough for whole cooperative waves (chunks 1 .. 64 m), with chunk 0 and
    the tail on the per-lane path beside them, over two calls (state and history carried)."""
    from tests.test_demod_gpu import _case
    spec, plain, x, ocfg, kw, fec, ofec = _case(case)
    need = int((75 * chunk + 12345) * (1.2 if case == "goes" else 1.0))  # (GOES: the stages run behind the 9/10 resampler)
    x = np.tile(x, need // len(x) + 1)[:need]
    out = {}
    for coop in ("0", "1"):
        monkeypatch.setenv("SDHIP_COOP", coop)
        monkeypatch.setenv("SDHIP_COOP_MM", coop)  # (the clock recovery's cooperative loads are off by default: measured slower; kept, tested here)
        if coop == "1":
            monkeypatch.setenv("SDHIP_COOP_REQUIRE", "1")  # the engine refuses to fall back silently: the first (long) call must run cooperative waves
        out[coop] = _run_demod(torch_cuda, capi, kw, x, chunks=[0, len(x) - 50000, len(x)], chunk_len=chunk)
    (s0, y0, st0), (s1, y1, st1) = out["0"], out["1"]
    assert st0.chunks == st1.chunks and st0.chunks_fixed == st1.chunks_fixed and st0.chunks_inexact == st1.chunks_inexact
    assert np.array_equal(s0, s1)
    assert np.array_equal(y0.view(np.uint32), y1.view(np.uint32))
