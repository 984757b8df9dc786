"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/sdhip.h declares
(no compute calls: there is no GPU here and no CPU fallback), and the host-side config plumbing works."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sdhip.h")).read()
    return sorted(set(re.findall(r"\b(sdhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from satdump_amd import capi
    L = capi.lib()
    missing = [s for s in _declared_symbols() if not hasattr(L, s)]
    assert not missing, f"declared in include/sdhip.h but not exported: {missing}"


def test_version_and_defaults():
    from satdump_amd import capi
    assert b"gfx950" in capi.lib().sdhip_version()
    c = capi.fec_cfg(constellation="qpsk", rs_i=4)
    assert c.constellation == capi.QPSK and c.asm_sync == 0x1ACFFC1D and c.rs_fill_bytes == -1 and c.derand_start == 4
    assert c.qpsk_swap_diff == 1 and c.conv_rate == capi.RATE_1_2 and c.device == 0


def test_struct_mirrors_match_the_header():
    """The ctypes mirrors (satdump_amd/capi.py and oracle/pyref.py) against include/sdhip.h: same field names in the same order
    (the structs are plain ints / floats / doubles, so order + count pins the layout)."""
    import re
    from satdump_amd import capi
    from oracle import pyref
    hdr = open(os.path.join(ROOT, "include", "sdhip.h")).read()

    def fields(name):
        body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for d in body.split(";"):
            if d.strip():
                names = d.strip().split(None, 1)[1] if not d.strip().startswith("unsigned") else d.strip().split(None, 2)[2]
                out += [re.sub(r"\[.*\]", "", n.strip()) for n in names.split(",")]
        return out

    assert [f for f, _ in capi.FecCfg._fields_] == fields("sdhip_fec_cfg")
    assert [f for f, _ in pyref.FecCfg._fields_] == fields("sdhip_fec_cfg")
    assert [f for f, _ in capi.DemodCfg._fields_] == fields("sdhip_demod_cfg")
    assert [f for f, _ in pyref.DemodCfg._fields_] == fields("sdhip_demod_cfg")
    assert [f for f, _ in capi.LrptCfg._fields_] == fields("sdhip_lrpt_cfg")
    assert [f for f, _ in capi.LrptStats._fields_] == fields("sdhip_lrpt_stats")
    assert [f for f, _ in capi.Dvbs2Stats._fields_] == fields("sdhip_dvbs2_stats")


def test_product_filter_design_against_the_reference_tables():
    """The PRODUCT's designers (satdump_amd/csrc/dsp_design.h, through sdhip_design of the real libsdhip.so -- host code, no device)
    against tests/golden/taps.npz, the tables the compiled reference built (tests/golden/make_golden.py): bit for bit."""
    import ctypes as C
    import numpy as np
    from satdump_amd import capi
    L = capi.lib()
    L.sdhip_design.restype = C.c_int64
    L.sdhip_design.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    d = np.load(os.path.join(ROOT, "tests", "golden", "taps.npz"))

    def design(kind, params):
        p = np.asarray(params, dtype=np.float64)
        out = np.zeros(1 << 16, dtype=np.float32)
        dims = np.zeros(3, dtype=np.int32)
        n = L.sdhip_design(kind, p.ctypes.data, out.ctypes.data, len(out), dims.ctypes.data)
        assert n > 0, capi.last_error()
        return out[:n], dims

    for key, fs, sr in (("rrc_goes", 2.7e6, 927000), ("rrc_metop", 6e6, 2333333)):
        t, _ = design(0, [1, fs, sr, 0.5, 31])
        assert np.array_equal(t.view(np.uint32), d[key].view(np.uint32)), key
    t, dims = design(1, [128, 8])
    assert list(dims[:2]) == [128, 8] and np.array_equal(t.reshape(128, 8).view(np.uint32), d["mm"].view(np.uint32))
    t, dims = design(2, [2700000, 3000000])
    assert list(dims[:2]) == list(d["resamp_ratio"]) and np.array_equal(t.reshape(dims[0], dims[2]).view(np.uint32), d["resamp"].view(np.uint32))
    # the pole of the RRC formula (4 alpha k / sps = 1) and the alpha = 1 branch evaluate
    t, _ = design(0, [1, 4.0, 1.0, 0.5, 33])
    assert np.all(np.isfinite(t)) and abs(float(t.sum()) - 1.0) < 1e-5 and np.array_equal(t, t[::-1])
    # and, where the compiled reference is at hand, a sweep of parameter sets against its own designers
    from oracle import pyref
    if pyref.ref_available():
        R = pyref.ref()
        for fs, sr, al, nt in [(4.0, 1.0, 0.5, 33), (4.0, 1.0, 1.0, 33), (8.0, 1.0, 0.25, 65), (30e6, 15e6, 0.35, 31), (2.5e6, 1.2e6, 0.6, 51), (6e6, 2333333, 0.2, 361)]:
            t, _ = design(0, [1, fs, sr, al, nt])
            assert np.array_equal(t.view(np.uint32), R.rrc_taps(fs, sr, al, nt).view(np.uint32)), (fs, sr, al, nt)
        for ip, dc in [(2700000, 3000000), (5, 7), (7, 5), (3, 4), (99, 100), (10, 17)]:
            t, dims = design(2, [ip, dc])
            bank, ir, dr = R.resamp_bank(ip, dc)
            assert [ir, dr] == list(dims[:2]) and np.array_equal(t.reshape(dims[0], dims[2]).view(np.uint32), bank.view(np.uint32)), (ip, dc)


def test_fails_loudly_without_gpu():
    import torch
    from satdump_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SdhipError):
        capi.FecDecoder(capi.fec_cfg(constellation="bpsk"))


def test_plugin_shim_builds_against_reference_headers():
    """plugin/sdhip_plugin.cpp (the ProcessingModule subclasses above the C ABI) compiles against the reference's own
    module.h / plugin.h and exports the plugin entry point `loader` (src-core/core/plugin.h:10-17); the only symbols it
    needs besides SatDump's are the C ABI's."""
    import subprocess
    if not os.path.isdir("/root/reference/src-core"):
        pytest.skip("reference tree not present (GPU box)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "plugin"), "all"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "plugin", "_build", "libsdhip_support.so")
    defined = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    assert " T loader" in defined
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", so], text=True)
    used = sorted(set(re.findall(r"\b(sdhip_[a-z0-9_]+)", undefined)))
    assert "sdhip_demod_push" in used and "sdhip_fec_push" in used and "sdhip_fec_pull" in used
    assert set(used) <= set(_declared_symbols())


def test_bench_reads_committed_pmc_traffic(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from the committed rocprofv3 --pmc summaries (profiles/r*_<wl>_pmc.csv), and only from a
    summary stamped with the hash of the kernel sources the library is built from: a stale profile yields traffic = None and says
    so. The algorithmic byte model must know every kernel the roofline may name."""
    import shutil
    import bench
    from satdump_amd import build as sd_build
    prof = tmp_path / "profiles"
    prof.mkdir()
    src = os.path.join(ROOT, "profiles", "history", "r01", "r01_j_goes_pmc.csv")  # an unstamped round-1 profile
    shutil.copy(src, prof / "r90_goes_pmc.csv")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    traffic, name = bench.pmc_traffic("goes_hrit", "k_mm")
    assert traffic is None and "stale" in name
    body = open(src).read().splitlines()
    (prof / "r91_goes_pmc.csv").write_text("\n".join([body[0], f"# source_hash: {sd_build.source_hash()}"] + body[1:]) + "\n")
    for k in ("k_mm", "k_chunks<AgcStage>", "k_chunks<CostasStage>", "k_vit2_acs"):
        traffic, name = bench.pmc_traffic("goes_hrit", k)
        assert name == "r91_goes_pmc.csv" and traffic and traffic > 1e6, (k, traffic, name)
    assert bench.pmc_traffic("goes_hrit", "k_chunks<AgcStage>")[0] != bench.pmc_traffic("goes_hrit", "k_chunks<CostasStage>")[0]
    wl = bench.WORKLOADS["goes_hrit"]
    algo = bench.algorithmic_bytes(wl, 262144000, 235929600, 81000000, 81000000, 4944 * 1024, 8)
    for k in ("k_mm", "k_chunks<AgcStage>", "k_chunks<CostasStage>", "k_resample", "k_resample_byoffset", "k_resample_period", "k_fir", "k_fir_window", "k_vit2_acs", "k_rs", "k_rs_screen"):
        assert algo[k] > 0
    # 8 + 2q/S + c/S bytes per input sample for GOES
    q, sps_in = wl["soft_per_sym"], wl["spec"]["samplerate"] / wl["spec"]["symbolrate"]
    assert abs(8 + 2 * q / sps_in + (q * wl["conv_rate"] / 8.0) / sps_in - 8.637) < 1e-3


def test_binding_refuses_the_host_twin(monkeypatch):
    """The host twin is for the test modules that open it explicitly; the binding does not take it as a backend."""
    import importlib.util
    from tests.emu import build as emu_build
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("no host clang++")
    monkeypatch.setenv("SDHIP_LIB", emu_build.build())
    monkeypatch.delenv("SDHIP_TESTING_TWIN", raising=False)
    spec = importlib.util.spec_from_file_location("capi_refuse", os.path.join(ROOT, "satdump_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    with pytest.raises(m.SdhipError):
        m.lib()
