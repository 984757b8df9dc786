"""The plugin of plugin/sdhip_plugin.cpp EXECUTED by a host built from the reference's own headers (tests/minihost): loader(),
init(), the RegisterModulesEvent / SatDumpStartedEvent handlers, the registry entries. No GPU here, so no process() call: the
override must notice that there is no HIP device and leave the CPU modules in place (tests/test_plugin_minihost_gpu.py runs
the modules)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "minihost", "_build", "minihost")
PLUGIN = os.path.join(ROOT, "plugin", "_build", "libsdhip_support.so")
LIB = os.path.join(ROOT, "satdump_amd", "lib", "libsdhip.so")


@pytest.fixture(scope="module")
def host():
    if os.path.isdir("/root/reference/src-core"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "plugin"), "all"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "minihost"), "all"], stdout=subprocess.DEVNULL)
    if not (os.path.exists(HOST) and os.path.exists(PLUGIN)):
        pytest.skip("minihost / plugin not built (needs the reference tree)")
    return HOST


def _run(host, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([host, LIB, PLUGIN, *args], capture_output=True, text=True, env=e, timeout=120)


def test_plugin_loads_and_registers_its_modules(host):
    p = _run(host, "list")
    assert p.returncode == 0, p.stderr
    ids = p.stdout.split()
    assert ids[0] == "sdhip_support"
    for m in ("psk_demod_hip", "ccsds_conv_concat_decoder_hip", "metop_ahrpt_decoder_hip", "ccsds_simple_psk_decoder_hip", "dvbs2_demod_hip"):
        assert m in ids[1:]
    # new ids are appended behind the core modules, the stock ids stay first (first-match lookup, module.cpp:123-129)
    assert ids.index("psk_demod") < ids.index("psk_demod_hip")


def test_override_without_a_device_leaves_the_cpu_modules(host, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    job = {"mode": "file", "input": str(tmp_path / "x.cf32"), "output_hint": str(tmp_path / "out"), "instantiate_only": True,
           "demod": {"module": "psk_demod", "parameters": {"samplerate": 6000000, "symbolrate": 2333333, "constellation": "qpsk", "rrc_alpha": 0.5, "pll_bw": 0.003}},
           "decoder": {"module": "metop_ahrpt_decoder", "parameters": {"viterbi_outsync_after": 10, "viterbi_ber_thresold": 0.28}}}
    jp = tmp_path / "job.json"
    jp.write_text(json.dumps(job))
    p = _run(host, "run", str(jp), env={"SDHIP_OVERRIDE": "1"})
    assert p.returncode == 0, p.stderr
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["demod_class"] == "cpu:psk_demod" and rep["decoder_class"] == "cpu:metop_ahrpt_decoder"
    # the explicit ids instantiate (constructor = parameter parsing only; init() is what needs the device)
    job["demod"]["module"], job["decoder"]["module"] = "psk_demod_hip", "metop_ahrpt_decoder_hip"
    jp.write_text(json.dumps(job))
    rep = json.loads(_run(host, "run", str(jp)).stdout.strip().splitlines()[-1])
    assert rep["demod_class"] == "psk_demod_hip" and rep["decoder_class"] == "metop_ahrpt_decoder_hip"
    # mandatory keys raise the reference's own messages
    del job["demod"]["parameters"]["pll_bw"]
    jp.write_text(json.dumps(job))
    p = _run(host, "run", str(jp))
    assert p.returncode != 0 and "PLL BW parameter must be present!" in p.stderr


def test_ndsp_block_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_ndsp_block_through_the_plugin with the host twin of the engine (tests/emu) as the C-ABI
    library: the ndsp::Block subclass of the plugin, its FIFO/terminator handling and set_cfg() mapping are exercised in the CPU suite."""
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.NdspRef.available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference ndsp blocks and a host clang++")
    G.check_ndsp_block_through_the_plugin(host, emu_build.build(), tmp_path)


def test_dvbs2_module_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_dvbs2_module_through_the_plugin with the host twin as the C-ABI library: DVBS2DemodHipModule's
    parameter parsing, table hand-over, file plumbing and statistics run in the CPU suite (short 8PSK 3/5 frames)."""
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not (pyref.Dvbs2Ref.available(False) and pyref.S2FrontRef.available()) or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference DVB-S2 classes and a host clang++")
    G.check_dvbs2_module_through_the_plugin(host, emu_build.build(), tmp_path, nfr=10, extra_legs=False)


def test_hip_devices_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_hip_devices_through_the_plugin with the host twin as the C-ABI library (its one "device" three times):
    the plugin's chunk threads, the alignment of the chunks' soft streams and the quarter-turn hand-over run in the CPU suite."""
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.ref_available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference and a host clang++")
    G.check_hip_devices_through_the_plugin(host, emu_build.build(), tmp_path, case="metop", nframes=60, devices=(0, 0), serial_chunks=True)


def test_ndsp_single_blocks_through_the_plugin_on_the_twin(host, tmp_path):
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.NdspRef.available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference ndsp blocks and a host clang++")
    G.check_ndsp_single_blocks_through_the_plugin(host, emu_build.build(), tmp_path)


def test_doppler_through_the_plugin_on_the_twin(host, tmp_path):
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.doppler_block_available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference Doppler block and a host clang++")
    G.check_doppler_through_the_plugin(host, emu_build.build(), tmp_path)


def test_lrpt_module_through_the_plugin_on_the_twin(host, tmp_path):
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_lrpt_decode")) or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference and a host clang++")
    G.check_lrpt_module_through_the_plugin(host, emu_build.build(), tmp_path, interleaved_run=bool(os.environ.get("SDHIP_TWIN_FULL")))  # (the interleaved run: ~10 min on the twin)


def test_fy3_module_through_the_plugin_on_the_twin(host, tmp_path):
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_decode")) or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference and a host clang++")
    G.check_fy3_module_through_the_plugin(host, emu_build.build(), tmp_path, variants=("short",))  # the end-of-file case; all three on the GPU


def test_fy3_mpt_module_through_the_plugin_on_the_twin(host, tmp_path):
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not (pyref.ref_available() and hasattr(pyref.ref().lib, "sdref_fy3_mpt_decode")) or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference and a host clang++")
    G.check_fy3_mpt_module_through_the_plugin(host, emu_build.build(), tmp_path)


def test_wav_container_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_wav_container_through_the_plugin with the host twin as the C-ABI library: the plugin's header detection, the
    `w16` / `wav` format names and the chunk plan behind a header run in the CPU suite."""
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("needs a host clang++")
    G.check_wav_container_through_the_plugin(host, emu_build.build(), tmp_path, nframes=10, serial_chunks=True)


def test_flowgraph_registry_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_flowgraph_registry_through_the_plugin with the host twin as the C-ABI library: the plugin built with
    -DSDHIP_WITH_FLOWGRAPH, its RegisterNodesEvent handler fired and its nodes made and run by the minihost -- in the CPU suite."""
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.NdspRef.available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference ndsp blocks and a host clang++")
    G.check_flowgraph_registry_through_the_plugin(host, emu_build.build(), tmp_path)


def test_decoder_hip_devices_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_decoder_hip_devices_through_the_plugin on the host twin (the chunks one after the other: the twin runs one kernel at
    a time): the concatenated decoder's case; MetOp and FengYun-3 stay with the GPU suite."""
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("no host clang++ to build the twin with")
    G.check_decoder_hip_devices_through_the_plugin(host, emu_build.build(), tmp_path, cases=("goes",), devices=(0, 0), serial_chunks=True)


def test_ziq_container_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_ziq_container_through_the_plugin with the host twin as the C-ABI library: the ZIQ header, the zstd stream undone
    through the system's libzstd and the chunk plan behind the header run in the CPU suite."""
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not os.path.exists(emu_build.CLANG):
        pytest.skip("needs a host clang++")
    G.check_ziq_container_through_the_plugin(host, emu_build.build(), tmp_path, nframes=10, serial_chunks=True,
                                             only=("bare16", "ziq16z", "bare8", "ziq8z", "bare16+devices", "ziq16+devices", "ziq16z+devices"))


def test_hard_symbols_through_the_plugin_on_the_twin(host, tmp_path):
    """tests/test_plugin_minihost_gpu.py::test_hard_symbols_through_the_plugin with the host twin as the C-ABI library: the plugin's packed-bit reader in front of
    ccsds_simple_psk_decoder, in the CPU suite."""
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.ref_available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference and a host clang++")
    G.check_hard_symbols_through_the_plugin(host, emu_build.build(), tmp_path, nframes=12)


def test_ts_extractor_through_the_plugin_on_the_twin(host, tmp_path):
    from oracle import pyref
    from tests import test_plugin_minihost_gpu as G
    from tests.emu import build as emu_build
    if not pyref.s2_ts_available() or not os.path.exists(emu_build.CLANG):
        pytest.skip("needs the compiled reference TS parser and a host clang++")
    G.check_ts_extractor_through_the_plugin(host, emu_build.build(), tmp_path)
