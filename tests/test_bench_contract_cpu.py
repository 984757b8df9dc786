"""What of bench.py's contract can be checked without a GPU: its next_rows tools import, the committed PMC profile belongs to the sources in the tree
(otherwise the driver line would carry traffic = null), the workload table and the algorithmic-byte figures of the dominant kernels."""
import os


def test_next_row_tools_import_and_parse():
    """bench.py imports tools/bench_ndsp.py and tools/bench_dvbs2.py for its next_rows object: they must import without a GPU and accept the argument
    lists bench.py hands them (a failure inside them is recorded in the JSON line, but a syntax error would be silly to find on the GPU box)."""
    import importlib
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        nd = importlib.import_module("bench_ndsp")
        s2 = importlib.import_module("bench_dvbs2")
        a = nd.parse(["--steps", "4", "--warmup", "1", "--cpu-samples", "12000000"])
        b = s2.parse(["--rate", "2/3", "--sigma", "13"])
        assert a.samples == 1 << 30 and a.constellation == "qpsk" and b.front == 1 and b.sync_frames == 128 and b.esn0 == 8.0
        assert callable(nd.run) and callable(s2.run)
        lr = importlib.import_module("bench_lrpt")
        assert lr.parse([]).frames == 8192 and callable(lr.run)
        fy = importlib.import_module("bench_fy3")
        assert fy.parse([]).frames == 98304 and callable(fy.run)
    finally:
        sys.path.remove(tools)


def test_committed_pmc_profile_belongs_to_the_sources_in_the_tree():
    """bench.py quotes roofline.traffic only from a PMC profile stamped with the hash of the kernel sources the library is built from. While kernels are
    being worked on the newest profile is stale by design (bench.py then reports traffic = null and says so): skipped with that message. Once a
    profile of the current sources is committed, it must yield a figure for the dominant lane kernels."""
    import pytest
    import bench
    from satdump_amd import build
    traffic, src = bench.pmc_traffic("metop_ahrpt", "k_afc")
    if src and "stale" in src and os.environ.get("SDHIP_FINAL") == "1":
        pytest.fail("SDHIP_FINAL=1 (the round's closing check): the committed PMC profile is of other kernel sources (" + src + ")")
    if src and "stale" in src:
        pytest.skip("the committed PMC profile is of other kernel sources (" + src + "): take the two --pmc passes again before the round ends")
    for k in ("k_afc", "k_mm", "k_vit2_acs", "k_compact8"):
        traffic, src = bench.pmc_traffic("metop_ahrpt", k)
        assert traffic and traffic > 1e9 and "stale" not in src, (k, traffic, src)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", src)) as f:
        assert f"# source_hash: {build.source_hash()}" in f.read()


def test_bench_workloads_and_algorithmic_bytes():
    """The three single-GPU workloads of BASELINE.json are there; the dominant kernels' algorithmic bytes are what DESIGN.md 4 states (16 B per resampled
    sample for the fused AGC + filter + Costas stage, 8 B in + 8 B per symbol out for the clock recovery)."""
    import bench
    assert set(bench.WORKLOADS) == {"goes_hrit", "metop_ahrpt", "npp_hrd"}
    wl = bench.WORKLOADS["metop_ahrpt"]
    a = bench.algorithmic_bytes(wl, 1000, 1000, 400, 800, 1024, 8)
    assert a["k_afc"] == 16000 and a["k_mm"] == 1000 * 8 + 400 * 8 and a["k_quantize"] == 400 * (8 + wl["soft_per_sym"])
    a = bench.algorithmic_bytes(wl, 1000, 1000, 400, 800, 1024, 8, q8=True)  # the timed steps: int8 symbols out of the clock recovery, compacted
    assert a["k_mm"] == 1000 * 8 + 400 * 2 and a["k_compact8"] == 400 * (2 + wl["soft_per_sym"])


def test_headline_line_is_compact_and_complete():
    """The driver keeps only the tail of stdout: round 5's one-line result had grown to 26.7 KB and BENCH_r05.json.parsed was null. bench.py now prints
    headline_line(out) -- built here from a committed full result object of a real run -- and writes the rest to bench_detail.json."""
    import glob
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cans = sorted(glob.glob(os.path.join(root, "profiles", "r0*_bench.json")) + glob.glob(os.path.join(root, "profiles", "r0*_bench_detail.json")))
    cans = [c for c in cans if os.path.getsize(c) > 8000]
    assert cans, "no committed full result object to build the line from"
    for can in cans[-3:]:
        with open(can) as f:
            out = json.load(f)
        line = bench.headline_line(out)
        assert "\n" not in line and len(line) < bench.HEADLINE_MAX_BYTES < 6000, (can, len(line))
        h = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                  "cpu_baseline"):
            assert k in h, k
        assert h["config"]["workload"] and "model" not in h["config"]
        assert h["value"] == out["value"] and h["ms_per_step"] == out["ms_per_step"]
        r = h["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algo_bytes_per_launch", "avg_launch_ms"):
            assert k in r, k
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        c = h["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert h["cadu_parity"]["byte_identical"] is True and h["soft_parity"]["frac_within_1e-5"] > 0.9
    # a pathological object (long strings everywhere) still yields a parseable line under the limit: optional parts are dropped
    out = dict(out, next_rows={f"row{i}": {"value": 1.0, "unit": "x" * 200} for i in range(40)})
    line = bench.headline_line(out)
    assert len(line) < bench.HEADLINE_MAX_BYTES and "roofline" in json.loads(line)
