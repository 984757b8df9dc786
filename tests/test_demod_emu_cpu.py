"""Host-logic tests of the demodulator engine WITHOUT a GPU (run with -m "not gpu").

The unchanged HIP sources satdump_amd/csrc/demod_{kernels,engine}.hip are compiled for the host against the stand-in runtime
in tests/emu (kernels run block by block, threads as fibers) and driven through the same C ABI as on the GPU. What this
covers: the engine's host side -- chunk geometry, speculation + boundary certificates + re-run rounds, Costas frame rotation,
M&M hand-off, state carry across calls -- and, because plain float arithmetic is IEEE on both sides, the kernels' arithmetic
as well. What it does NOT cover: anything wave-level or timing related, and the FEC engine (its kernels are written in
gfx950 instructions). The product never loads this library; the GPU parity tests (tests/test_demod_gpu.py) stay the gate."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth
from tests import util
from tests.emu import build as emu_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_TOL = 1e-5


@pytest.fixture(scope="module")
def twin():
    """satdump_amd/capi.py bound to the host twin (a second module object: the real binding stays untouched)."""
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("no host clang++ to build the twin with")
    lib = emu_build.build()
    spec = importlib.util.spec_from_file_location("capi_host_twin", os.path.join(ROOT, "satdump_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    old = os.environ.get("SDHIP_LIB")
    os.environ["SDHIP_LIB"] = lib
    os.environ["SDHIP_TESTING_TWIN"] = "1"  # capi refuses the twin without it
    try:
        spec.loader.exec_module(m)
        m.lib()
    finally:
        del os.environ["SDHIP_TESTING_TWIN"]
        if old is None:
            del os.environ["SDHIP_LIB"]
        else:
            os.environ["SDHIP_LIB"] = old
    assert m.LIB_PATH == lib  # the twin, not lib/libsdhip.so
    return m


@pytest.fixture(scope="module")
def orc():
    return pyref.best()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p).value


BLOCKS = [(0, [1e-2, 1, 1, 65536]), (1, [6e6, 2333333, 0.5, 31]), (2, [0.003, 4, 1.0]), (2, [0.02, 2, 1.0]), (2, [0.003, 8, 1.0]),
          (3, [2.5714, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005]), (4, [2700000, 3000000]), (4, [5, 7]), (5, [0.0]),
          (7, [2.5714, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005])]


@pytest.mark.parametrize("kind,params", BLOCKS)
def test_single_blocks_bit_exact(twin, orc, kind, params):
    rng = np.random.default_rng(kind + 11)
    n = 30011  # ragged: last tiles of the window kernels are partial
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
    want = orc.block(kind, params, x)
    y = np.zeros(2 * (n + 64), dtype=np.float32)
    p = np.asarray(params, dtype=np.float32)
    nout = twin.lib().sdhip_op_block(0, kind, p.ctypes.data_as(C.c_void_p), C.c_void_p(_ptr(x)), n, C.c_void_p(_ptr(y)), n + 64)
    assert nout >= 0, twin.last_error()
    got = y[: 2 * nout].view(np.complex64)
    assert len(got) == len(want)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _case(name, nframes):
    if name == "goes":
        spec, cadus, plain, syms = util.goes_case(nframes=nframes)
        ocfg = pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0)
        kw = dict(samplerate=3e6, symbolrate=927000, constellation="bpsk", rrc_alpha=0.5, pll_bw=0.02, max_sps=3.0)
        ofec = pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=1)
    elif name == "metop":
        spec, cadus, plain, syms = util.metop_case(nframes=nframes)
        ocfg = pyref.demod_cfg()
        kw = dict(samplerate=6e6, symbolrate=2333333, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.003)
        ofec = None
    else:
        spec, cadus, plain, syms = util.npp_case(nframes=nframes)
        ocfg = pyref.demod_cfg(samplerate=30e6, symbolrate=15e6, constellation=pyref.QPSK, pll_bw=0.002)
        kw = dict(samplerate=30e6, symbolrate=15e6, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.002)
        ofec = pyref.fec_cfg(constellation=pyref.QPSK, nrzm=1, rs_usecheck=1)
    x, _ = synth.modulate(syms, spec)
    return plain, x, ocfg, kw, ofec


def _run(twin, kw, x, chunks=None, **extra):
    cfg = twin.demod_cfg(**kw, **extra)
    dem = twin.PskDemod(cfg)
    x = np.ascontiguousarray(x)
    soft, syms = [], []
    bounds = chunks or [0, len(x)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        n = b - a
        o_soft = np.zeros(2 * n + 64, dtype=np.int8)
        o_syms = np.zeros(2 * (n + 64), dtype=np.float32)
        ns = dem.process_dev(_ptr(x) + 8 * a, n, twin.FMT_CF32, _ptr(o_soft), 2 * n + 64, _ptr(o_syms), n + 64)
        nsym = ns if cfg.constellation == twin.BPSK else ns // 2
        soft.append(o_soft[:ns].copy())
        syms.append(o_syms[: 2 * nsym].view(np.complex64).copy())
    st = dem.stats()
    dem.close()
    return np.concatenate(soft), np.concatenate(syms), st


def _cadus(orc, case, ofec, soft):
    return orc.metop_decode(soft)["cadu"] if case == "metop" else orc.concat_decode(ofec, soft)["cadu"]


@pytest.mark.parametrize("case", ["goes", "metop", "npp"])
def test_exact_mode_streaming_bit_identical(twin, orc, case):
    plain, x, ocfg, kw, ofec = _case(case, 6)
    x = x[:150000]
    want = orc.psk_demod(ocfg, x)
    soft, syms, st = _run(twin, kw, x, chunks=[0, 0, 7, 1000, 1001, 77777, len(x)], exact=1)
    assert st.buffer_size == want["buffer_size"] and st.final_sps == np.float32(want["final_sps"])
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32))
    assert np.array_equal(soft, want["soft"])


@pytest.mark.parametrize("case,chunk", [("goes", 8192), ("metop", 8192), ("npp", 4096)])
def test_chunked_mode_against_reference(twin, orc, case, chunk):
    """The chunk-speculative engine on the host twin: same symbol count, float symbols within the contract, CADUs (decoded by the
    reference's FEC from either soft stream) identical; several calls, so state, history and Costas frame carry are on the path."""
    plain, x, ocfg, kw, ofec = _case(case, 24 if case == "goes" else 40)
    want = orc.psk_demod(ocfg, x)
    n = len(x)
    soft, syms, st = _run(twin, kw, x, chunks=[0, n // 3, n // 3 + 12345, n], chunk_len=chunk)
    assert st.chunks > 30 and st.chunks_forced == 0
    assert len(syms) == len(want["syms"])
    ref = want["syms"]
    err = np.abs(syms - ref) / np.sqrt(np.mean(np.abs(ref) ** 2))
    assert np.median(err) < 1e-6
    assert np.mean(err > REL_TOL) < (0.012 if case == "goes" else 0.007)  # measured 0.006-0.008 / 0.003-0.004 (tools/twin/soft_parity.py)
    assert err.max() < 0.08
    d = soft.astype(np.int32) - want["soft"].astype(np.int32)
    assert np.abs(d).max() <= 4 and np.mean(d != 0) < 0.004
    assert st.chunks_fixed <= st.chunks // 10
    got, wantc = _cadus(orc, case, ofec, soft), _cadus(orc, case, ofec, want["soft"])
    assert len(wantc) >= 8
    assert got.shape == wantc.shape and np.array_equal(got, wantc)


def test_rerun_rounds_are_exercised(twin, orc, monkeypatch):
    """Warm-ups cut to a fraction of what the loops need: many boundaries miss their certificate, the engine re-runs those chunks from
    the exact predecessor states (several rounds) -- and the result still meets the contract."""
    plain, x, ocfg, kw, ofec = _case("npp", 40)
    want = orc.psk_demod(ocfg, x)
    monkeypatch.setenv("SDHIP_W_MM", "256")
    monkeypatch.setenv("SDHIP_W_COSTAS", "256")
    monkeypatch.setenv("SDHIP_W_AGC", "256")
    soft, syms, st = _run(twin, kw, x, chunk_len=4096)
    assert st.chunks_fixed > 0
    assert len(syms) == len(want["syms"])
    got, wantc = _cadus(orc, "npp", ofec, soft), _cadus(orc, "npp", ofec, want["soft"])
    assert got.shape == wantc.shape and np.array_equal(got, wantc)


def test_auto_geometry_per_stage(twin, orc, monkeypatch):
    """pick_L with different lane targets per stage (the chunk grids of AGC, Costas and M&M no longer coincide)."""
    plain, x, ocfg, kw, ofec = _case("metop", 40)
    want = orc.psk_demod(ocfg, x)
    monkeypatch.setenv("SDHIP_LANES_AGC", "40")
    monkeypatch.setenv("SDHIP_LANES_COSTAS", "150")
    monkeypatch.setenv("SDHIP_LANES_MM", "64")
    soft, syms, st = _run(twin, kw, x)
    assert st.chunks > 100
    assert len(syms) == len(want["syms"])
    got, wantc = _cadus(orc, "metop", None, soft), _cadus(orc, "metop", None, want["soft"])
    assert got.shape == wantc.shape and np.array_equal(got, wantc)


def _noise(rng, m, sig):
    return ((rng.standard_normal(m) + 1j * rng.standard_normal(m)) * sig / np.sqrt(2)).astype(np.complex64)


@pytest.mark.parametrize("scenario", ["noise_first", "amplitude_step", "noise_gap", "frequency_step"])
def test_disturbed_streams_deliver_the_reference_frames(twin, orc, scenario):
    """Streams that are not stationary (NPP: the narrowest carrier loop of the three configs, so the one that depends most on the
    warm-up start values). Found with this twin and fixed: a start-frequency estimate weighted by |x|^order let the AGC transient
    after a level step, or 100 k samples of leading noise, throw every chunk's warm-up off (0 of 48 / 58 frames).
    Where the loops are unlocked both decoders deliver nothing, so the frame sets are compared."""
    plain, x, ocfg, kw, ofec = _case("npp", 60)
    n, sig = len(x), float(np.std(x))
    rng = np.random.default_rng(4)
    if scenario == "noise_first":
        x = np.concatenate([_noise(rng, 100000, sig), x])
    elif scenario == "amplitude_step":
        x = x.copy()
        x[n // 3: 2 * n // 3] *= 4
    elif scenario == "noise_gap":
        x = np.concatenate([x[: n // 2], _noise(rng, 150000, sig), x[n // 2:]])
    else:
        x = x.copy()
        x[n // 2:] *= np.exp(1j * 0.002 * np.arange(n - n // 2)).astype(np.complex64)
    want = orc.psk_demod(ocfg, x)
    soft, syms, st = _run(twin, kw, x, chunk_len=4096)
    got, wantc = _cadus(orc, "npp", ofec, soft), _cadus(orc, "npp", ofec, want["soft"])
    ws, gs = {bytes(c) for c in wantc}, {bytes(c) for c in got}
    assert len(ws) >= 40
    # around an unlocked stretch the two re-acquire at their own pace (there is no sequential trajectory to be faithful to there)
    slack = 3 if scenario in ("noise_first", "noise_gap") else 0
    assert len(ws - gs) <= slack and len(gs - ws) <= slack
    if scenario in ("amplitude_step", "frequency_step"):
        assert st.chunks_forced == 0 and len(syms) == len(want["syms"])


@pytest.mark.parametrize("seed", range(10))
def test_random_configurations_exact_mode(twin, orc, seed):
    """Differential fuzz: random sample/symbol rates (with and without the rational resampler), tap counts, loop gains, dc_block,
    iq_swap and call boundaries -- exact mode must reproduce the reference chain bit for bit. (How dc_block + resampler reading
    the wrong buffer was found.)"""
    rng = np.random.default_rng(1000 + seed)
    const = str(rng.choice(["bpsk", "qpsk", "qpsk", "oqpsk", "8psk"]))
    symrate = float(rng.choice([927000, 2333333, 665400, 3.5e6]))
    fs = round(symrate * float(rng.uniform(1.3, 7.0)) / 1000) * 1000.0
    kw = dict(samplerate=fs, symbolrate=symrate, rrc_alpha=float(rng.choice([0.35, 0.5, 0.6])), rrc_taps=int(rng.choice([31, 51, 21])),
              pll_bw=float(rng.choice([0.002, 0.006, 0.02])), agc_rate=float(rng.choice([1e-2, 1e-3])), dc_block=int(seed % 3 == 0), iq_swap=int(seed % 4 == 1))
    n = int(rng.integers(20000, 60000))
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
    cn = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK, "oqpsk": pyref.OQPSK, "8psk": pyref.PSK8}[const]
    try:
        want = orc.psk_demod(pyref.demod_cfg(constellation=cn, **kw), x)
    except Exception:
        pytest.skip("the oracle refuses this parameter set")
    cuts = sorted(set([0, n] + rng.integers(0, n, 3).tolist()))
    try:
        soft, syms, st = _run(twin, dict(constellation=const, **kw), x, chunks=cuts, exact=1)
    except Exception as e:
        if "longer than the history" in str(e):
            pytest.skip("resampler bank longer than the 64-sample history window (engine refuses, plugin keeps the CPU module)")
        raise
    assert st.buffer_size == want["buffer_size"]
    assert np.array_equal(syms.view(np.uint32), want["syms"].view(np.uint32))
    assert np.array_equal(soft, want["soft"])


@pytest.mark.parametrize("case,chunk,env", [
    ("npp", 2048, {"SDHIP_CKPT": "1", "SDHIP_W_MM": "512"}),
    ("npp", 4096, {"SDHIP_CKPT": "1", "SDHIP_W_COSTAS": "512", "SDHIP_W_AGC": "256"}),
    ("goes", 4096, {"SDHIP_CKPT": "1", "SDHIP_W_COSTAS": "256", "SDHIP_W_MM": "512"}),
    ("metop", 8192, {"SDHIP_CKPT": "1"}),
])
def test_experimental_early_exit_of_rerun_lanes(twin, orc, monkeypatch, case, chunk, env):
    """SDHIP_CKPT=1 (off by default, not yet validated on the GPU): re-run lanes stop at the first checkpoint where they have merged
    with the earlier run of their chunk. With warm-ups cut so that many boundaries fail: still every symbol, still the reference's
    frames. (The first version lost a symbol now and then: a stale checkpoint of an earlier call behind the last one of this call.)"""
    plain, x, ocfg, kw, ofec = _case(case, 50)
    want = orc.psk_demod(ocfg, x)
    n = len(x)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    soft, syms, st = _run(twin, kw, x, chunks=[0, (2 * n) // 3 + 99, (4 * n) // 5 + 55, n], chunk_len=chunk)
    assert len(syms) == len(want["syms"])
    got, wantc = _cadus(orc, case, ofec, soft), _cadus(orc, case, ofec, want["soft"])
    assert len(wantc) >= 30 and got.shape == wantc.shape and np.array_equal(got, wantc)


@pytest.mark.parametrize("case,extra,env", [("goes", dict(chunk_len=4096), {}), ("npp", dict(chunk_len=2048), {"SDHIP_W_MM": "512"}), ("metop", dict(exact=1), {})])
def test_experimental_split_symbol_loop_is_bit_identical(twin, monkeypatch, case, extra, env):
    """SDHIP_MM_SPLIT=1 (off by default, not yet measured): the M&M symbol loop as one plain loop per phase. Same soft and float symbols, bit for bit."""
    plain, x, ocfg, kw, ofec = _case(case, 24)
    n = len(x)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    a = _run(twin, kw, x, chunks=[0, n // 3 + 5, n], **extra)
    monkeypatch.setenv("SDHIP_MM_SPLIT", "1")
    b = _run(twin, kw, x, chunks=[0, n // 3 + 5, n], **extra)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
