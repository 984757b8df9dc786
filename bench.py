#!/usr/bin/env python3
"""bench.py -- Msamples/s of IQ through psk_demod -> Viterbi -> deframe -> derand -> RS on MI355X.

One "step" = one pass of the whole hot path (sdhip_demod_process_dev + sdhip_fec_process_dev, include/sdhip.h) over
one batch of synthetic IQ that is ALREADY RESIDENT IN HBM when the timed region starts; soft symbols and CADUs stay in
HBM too. Default workload = BASELINE.json configs[2], the largest single-GPU configuration (MetOp AHRPT: QPSK 2.33 Msym/s @
6 Msps cf32, punctured r=3/4 Viterbi + RS(255,223) I=4, 16 GiB of IQ); --workload selects goes_hrit (configs[1]) or
npp_hrd (configs[3]'s per-GPU share).

N=1: the engines are stateful and the recording tiles seamlessly in time, so consecutive steps are one continuous stream.
N>1 (torchrun, one rank per GPU): ONE recording of N x (the per-GPU share) samples is cut into contiguous per-rank chunks
(satdump_amd/shard.py: rank r reads from `overlap` samples before its range so that its loops, Viterbi and deframer are locked
when its own range begins); every step each rank starts cold (fresh handles), demodulates and decodes its chunk, and rank 0
stitches the per-rank CADU lists on the host from the boundary frames (no data-path collective); value =
samples of the recording / max-over-ranks time, scaling "weak" (the share per GPU is fixed).

Prints ONE compact JSON line on rank 0 (headline(): < 4 KB, no prose) and writes the full result object to bench_detail.json. The full object's extra parts: "roofline" (dominant kernel, HIP-event timed on the launch stream via
sdhip_prof_*), "cpu_baseline" (the compiled reference oracle/_ref in its own thread-per-block topology over the first
--parity-samples samples of the stream -- default: ALL of it -- plus single-thread and all-cores legs on a bounded sample; rank 0 /
N=1 only), "cadu_parity" (the first pass of fresh handles over the FULL-SIZE stream against that reference run: every CADU the
reference produced, the RS-uncorrectable ones included, must be byte-identical -- hard failure otherwise), "soft_parity" (int8
soft symbols over the same span, float symbols over the first --cpu-samples samples), "cadu_per_s", "kernels" (per-kernel
ms/step), "other_workloads": the same measurement, compact, for the other two single-GPU workloads (--others 0 to skip), and
"next_rows": the rows SURVEY.md 8 marks "next" that have a measurement of their own -- the ndsp PSK demodulator chain
(tools/bench_ndsp.py), the DVB-S2 FEC tail (tools/bench_dvbs2.py) and BASELINE configs[4] itself, 8PSK baseband -> BBFRAMEs through the module-shaped
handle (tools/bench_dvbs2_demod.py), the Meteor LRPT decoder (tools/bench_lrpt.py), the FengYun-3 AHRPT decoder (tools/bench_fy3.py) -- as those tools report them (--next-rows 0 to skip).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # BASELINE.json configs[1]: GOES HRIT BPSK + Viterbi r=1/2 + RS(255,223), 2 GB cf32 @ 3 Msps
    "goes_hrit": dict(
        spec=dict(constellation="bpsk", samplerate=3e6, symbolrate=927e3, conv="1/2", nrzm=True, esn0_db=7.0, amplitude=0.5, cfo_hz=1000.0, seed=2),
        demod=dict(samplerate=3e6, symbolrate=927000, constellation="bpsk", rrc_alpha=0.5, pll_bw=0.02, max_sps=3.0),
        fec=dict(constellation="bpsk", cadu_size=8192, viterbi_ber_thresold=0.3, viterbi_outsync_after=20, derandomize=1, nrzm=1, rs_i=4,
                 rs_type=1, rs_usecheck=1),
        frames_quantum=309, frames=16 * 309, q=1, soft_per_sym=1, conv_rate=0.5, baseline="configs[1]"),
    # BASELINE.json configs[2]: MetOp AHRPT QPSK + punctured r=3/4 + RS, 16 GB cf32 @ 6 Msps
    "metop_ahrpt": dict(
        spec=dict(constellation="qpsk", samplerate=6e6, symbolrate=2333333, conv="3/4-metop", nrzm=False, esn0_db=10.0, amplitude=0.25,
                  cfo_hz=3000.0, seed=3),
        demod=dict(samplerate=6e6, symbolrate=2333333, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.003),
        fec=dict(decoder=1, viterbi_ber_thresold=0.28, viterbi_outsync_after=10),
        frames_quantum=21, frames=21 * 7280, q=2, soft_per_sym=2, conv_rate=0.75, baseline="configs[2]"),
    # BASELINE.json configs[3] per-GPU share: JPSS HRD QPSK 15 Msym/s @ 30 Msps, 16 GB cf32 per GPU
    "npp_hrd": dict(
        spec=dict(constellation="qpsk", samplerate=30e6, symbolrate=15e6, conv="1/2", nrzm=True, esn0_db=8.0, amplitude=0.4, cfo_hz=20000.0,
                  seed=4),
        demod=dict(samplerate=30e6, symbolrate=15e6, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.002),
        fec=dict(constellation="qpsk", cadu_size=8192, viterbi_ber_thresold=0.3, viterbi_outsync_after=20, derandomize=1, nrzm=1, rs_i=4,
                 rs_type=1, rs_usecheck=1),
        frames_quantum=1, frames=131072, q=2, soft_per_sym=2, conv_rate=0.5, baseline="configs[3] (per-GPU share)"),
}


def algorithmic_bytes(wl, n_in, n_rs, nsym, nsoft, cadu_bytes_out, in_bytes_per_sample, q8=False):
    """Per-step minimal HBM traffic of each kernel when stages are NOT fused: read + write once. q8: the clock recovery
    writes the module's int8 soft symbols (two bytes per symbol in its rows, k_compact8 behind it) instead of float symbols (eight bytes,
    k_quantize behind it) -- what it does whenever nobody asks for the float symbols, i.e. in the timed steps."""
    q = wl["soft_per_sym"]
    return {
        "k_convert": n_in * (in_bytes_per_sample + 8),
        "k_resample": n_in * 8 + n_rs * 8,
        "k_resample_byoffset": n_in * 8 + n_rs * 8,
        "k_resample_period": n_in * 8 + n_rs * 8,
        "k_chunks<AgcStage>": n_rs * 16,
        "k_chunks<AgcFirStage>": n_rs * 16,  # AGC and the RRC filter in one pass: the AGC samples never reach memory
        "k_chunks<PllStage>": n_rs * 16,
        "k_fir": n_rs * 16,
        "k_fir_window": n_rs * 16,
        "k_chunks<CostasStage>": n_rs * 16,
        "k_afc": n_rs * 16,  # AGC + RRC filter + Costas loop in one pass: neither the AGC nor the filtered samples reach memory
        "k_mm": n_rs * 8 + nsym * (2 if q8 else 8),
        "k_quantize": nsym * (8 + q),
        "k_compact8": nsym * (2 + q),
        "k_vit_decode": nsoft + nsoft * wl["conv_rate"] / 8.0,
        "k_vit2_acs": nsoft + nsoft * wl["conv_rate"] / 8.0,  # the Viterbi stage as a whole (prep/acs/tb/cert); SURVEY 8(d) keeps decisions on chip
        "k_vit_ber": nsoft + nsoft * wl["conv_rate"] / 8.0,
        "k_sync_search": nsoft * wl["conv_rate"] / 8.0,
        "k_pack_stream": 2 * nsoft * wl["conv_rate"] / 8.0,
        "k_extract": 2 * cadu_bytes_out,
        "k_rs": 2 * cadu_bytes_out,
        "k_rs_screen": cadu_bytes_out,
        "k_compact": 2 * cadu_bytes_out,
    }


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r*_<wl>_pmc.csv,
    produced by tools/gpu_visit.sh's pmc stage + tools/pmc_summary.py with the guide's gfx950 x2 correction on FETCH_SIZE). The PMC passes cannot
    run inside this process, so the figure is read from the most recent committed profile of this workload -- and only if that
    profile was taken with the kernel sources this library was built from (its `# source_hash:` line against
    satdump_amd.build.source_hash()); a profile of other sources yields traffic = null and says so in traffic_source."""
    import csv
    import glob
    from satdump_amd import build as sd_build
    short = {"goes_hrit": "goes", "metop_ahrpt": "metop", "npp_hrd": "npp"}.get(workload, workload)  # (the next-row benches pass their own tag: dvbs2, dvbs2fec, lrpt, fy3, ndsp)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{short}_pmc.csv")))
    if not files:
        return None, None
    name = os.path.basename(files[-1])
    with open(files[-1]) as f:
        lines = f.read().splitlines()
    prof_hash = next((ln.split(":", 1)[1].strip() for ln in lines if ln.startswith("# source_hash:")), None)
    if prof_hash != sd_build.source_hash():
        return None, f"{name} (stale: taken with kernel sources {prof_hash}, this library is {sd_build.source_hash()})"
    rows = [r for r in csv.reader(ln for ln in lines if not ln.startswith("#"))]
    hdr, rows = rows[0], rows[1:]
    # the engine times the Viterbi lanes under the stage names k_vit2_acs / k_vit2_tb; the symbols rocprofv3 sees since round 4 are the path-history
    # kernels k_vit2h_acs / k_vit2h_tb (the decision-word kernels keep the plain names: SDHIP_VIT2_HIST=0)
    alias = {"k_vit2_acs": ("k_vit2_acs", "k_vit2h_acs"), "k_vit2_tb": ("k_vit2_tb", "k_vit2h_tb")}.get(kernel.split("<")[0], (kernel.split("<")[0],))
    for r in rows:
        if r and r[0].split("<")[0] in alias and (("<" not in kernel) or kernel.split("<")[1].split(">")[0].split(",")[0] in r[0]):
            return float(r[-1]), name  # traffic_bytes_per_dispatch is the last column (kernel names carry commas: count from the right)
    return None, name


def pmc_traffic_per_step(workload, kernel, launches_per_step):
    """pmc_traffic's bytes per DISPATCH brought to the basis the algorithmic bytes of the row tools are quoted on -- per STEP: x the kernel's launches per step
    (the two were printed side by side). Returns (bytes per step, bytes per launch, source)."""
    tr, src = pmc_traffic(workload, kernel)
    if tr is None:
        return None, None, src
    return tr * launches_per_step, tr, src


def ref_cfgs(wl):
    from oracle import pyref
    d, f = wl["demod"], wl["fec"]
    cons = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK}[d["constellation"]]
    ocfg = pyref.demod_cfg(samplerate=d["samplerate"], symbolrate=d["symbolrate"], constellation=cons, rrc_alpha=d["rrc_alpha"], pll_bw=d["pll_bw"],
                           max_sps=d.get("max_sps", 4.0))
    metop = f.get("decoder", 0) == 1
    if metop:
        ofec = pyref.fec_cfg(viterbi_ber_thresold=f["viterbi_ber_thresold"], viterbi_outsync_after=f["viterbi_outsync_after"])
    else:
        ofec = pyref.fec_cfg(constellation=cons, nrzm=f["nrzm"], rs_usecheck=f["rs_usecheck"], viterbi_ber_thresold=f["viterbi_ber_thresold"],
                             viterbi_outsync_after=f["viterbi_outsync_after"])
    return ocfg, ofec, metop


def ref_decode(orc, wl, x_host, want_syms):
    ocfg, ofec, metop = ref_cfgs(wl)
    t0 = time.perf_counter()
    r = orc.psk_demod(ocfg, x_host, want_syms=want_syms)
    t1 = time.perf_counter()
    if metop:
        out = orc.metop_decode(r["soft"], ber_thr=ofec.viterbi_ber_thresold, outsync_after=ofec.viterbi_outsync_after)
    else:
        out = orc.concat_decode(ofec, r["soft"])
    t2 = time.perf_counter()
    return r, out["cadu"], t1 - t0, t2 - t1


def cpu_baseline(wl, x_host, n_prefix):
    """The reference's own code (oracle/_ref; the restatement if it is not there) on the same stream:
    (i) in the reference's run-time topology -- a thread per DSP block, the module thread, the decoder module's thread
        (pipeline_run.cpp:72-104, block.h:49-53) -- over ALL of x_host, which is `value` and whose CADUs / soft symbols are what the
        parity legs compare against;
    (ii) one thread (the same calls back to back) on the first n_prefix samples; its float symbols feed soft_parity;
    (iii) all host cores: one independent reference instance per core on its own slice of the prefix."""
    import threading
    from oracle import pyref
    kind = "reference" if pyref.ref_available() else "port"
    orc = pyref.best()
    n = len(x_host)
    n_prefix = min(n, n_prefix)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    r, cadus, t_dem, t_fec = ref_decode(orc, wl, x_host[:n_prefix], want_syms=True)
    single = {"value": round(n_prefix / (t_dem + t_fec) / 1e6, 3), "cores": 1, "demod_s": round(t_dem, 2), "fec_s": round(t_fec, 2),
              "sample": f"first {n_prefix} samples"}
    res = {"unit": "Msamples/s", "kind": kind, "host_cores": ncores, "single_thread": single,
           "volk": "generic-order shim (ref_shim/), gcc -O2, no libvolk on the box: a pessimistic baseline (a tuned libvolk would be faster)"}
    full = {"cadu": cadus, "soft": r["soft"], "samples": n_prefix}
    if kind == "reference":
        ocfg, ofec, metop = ref_cfgs(wl)
        th = pyref.ref().pipeline_threaded(ocfg, ofec, 1 if metop else 0, x_host, keep_soft=True)
        res.update({"value": round(n / th["seconds"] / 1e6, 3), "cores": int(th["threads"]),
                    "sample": f"first {n} samples of rank 0's stream; reference topology: {th['threads']} threads (one per DSP block + source + module + decoder), "
                              f"{len(th['cadu'])} CADUs in {th['seconds']:.2f} s"})
        full = {"cadu": th["cadu"], "soft": th["soft"], "samples": n}
    else:
        res.update({"value": single["value"], "cores": 1, "sample": f"first {n_prefix} samples of rank 0's stream, one thread"})
    # all cores: one independent reference instance per host core, each on its own 2 M-sample window of the prefix (the windows
    # are spread evenly over it and overlap when there are more cores than prefix / 2 M: a throughput figure, instance i has
    # nothing to do with instance j)
    per = min(n_prefix, 2_000_000)
    k = ncores
    if k > 1 and per >= 500_000:
        starts = [(i * (n_prefix - per)) // max(1, k - 1) for i in range(k)]

        def work(i):
            ref_decode(orc, wl, x_host[starts[i]:starts[i] + per], want_syms=False)

        ths = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        res["all_cores"] = {"value": round(k * per / dt / 1e6, 3), "cores": k, "sample": f"{k} independent instances x {per} samples in {dt:.2f} s"}
    return res, r, full


# What the chunk-parallel mode is held to at bench size (first pass of fresh handles over the full-size stream) -- the arm-grid contract, per workload
# class: every CADU byte-identical (checked apart, hard); of the float symbols of the prefix: at most `other` of them interpolated two or more grid steps from the
# reference's position (runs behind a slicer disagreement of the timing detector), at most `flip` one step off (the arm flicker), at most `same` of the symbols ON the
# reference's arm beyond 1e-5 (runs behind a sign-detector disagreement of the carrier loop); of the int8 stream of the timed handles: at least `eq` equal, none more
# than `lsb` apart, and the float instantiation's int8 stream the same bytes. Measured round 6 (profiles/r06_*): MetOp other 1.6e-5 / flip 3.5e-3 / same 3.3e-4 /
# eq 0.99915 / 5 LSB; GOES 4.7e-4 / 7.4e-3 / 5.5e-4 / 0.9989 / 3; NPP 1.1e-4 / 4.2e-3 / 8.3e-4 / 0.9986 / 7. A line outside these fails (exit code 3).
PARITY_GATES = {"goes_hrit": dict(other=1e-3, flip=0.012, same=1.2e-3, eq=0.998, lsb=6), "metop_ahrpt": dict(other=1e-4, flip=0.008, same=8e-4, eq=0.9985, lsb=10),
                "npp_hrd": dict(other=4e-4, flip=0.008, same=1.6e-3, eq=0.998, lsb=12)}


def parity_gates(workload, sp):
    g = PARITY_GATES[workload]
    ag = sp.get("arm_grid")
    res = {"limits": g}
    ok = sp["frac_int8_equal"] >= g["eq"] and sp["max_lsb"] <= g["lsb"]
    res["int8"] = {"frac_equal": sp["frac_int8_equal"], "max_lsb": sp["max_lsb"]}
    if sp.get("float_path") is not None:
        res["float_instantiation_same_bytes"] = bool(sp["float_path"]["identical_to_the_timed_handles_stream"])
        ok = ok and res["float_instantiation_same_bytes"]
    if ag:
        n = max(1, ag["symbols"])
        res["arm_grid"] = {"other_frac": round(ag["other"] / n, 7), "one_arm_step_frac": ag["one_arm_step"]["frac"],
                           "same_arm_beyond_1e-5_frac": round(ag["same_arm"]["beyond_1e-5"] / n, 7)}
        ok = ok and ag["other"] / n <= g["other"] and ag["one_arm_step"]["frac"] <= g["flip"] and ag["same_arm"]["beyond_1e-5"] / n <= g["same"]
        ok = ok and bool(ag["restatement_symbols_bit_identical_to_the_reference"])
    res["passed"] = bool(ok)
    return res


def int8_parity(gpu_soft, ref_soft_full):
    m = min(len(ref_soft_full), len(gpu_soft))
    d = np.zeros(0, dtype=np.int16)
    hist = np.zeros(6, dtype=np.int64)
    mx = 0
    for a in range(0, m, 1 << 28):  # in pieces: the streams are gigabytes
        d = np.abs(gpu_soft[a:min(m, a + (1 << 28))].astype(np.int16) - ref_soft_full[a:min(m, a + (1 << 28))].astype(np.int16))
        hist += np.bincount(np.minimum(d, 5).astype(np.int64), minlength=6)
        mx = max(mx, int(d.max()) if len(d) else 0)
    return {"int8_compared": int(m), "frac_int8_equal": round(float(hist[0] / max(1, m)), 6),
            "int8_abs_diff_hist": {"0": int(hist[0]), "1": int(hist[1]), "2": int(hist[2]), "3": int(hist[3]), "4": int(hist[4]), ">=5": int(hist[5])},
            "max_lsb": mx}


def soft_parity(gpu_syms, gpu_soft, gpu_soft_float_path, ref, ref_soft_full):
    """Agreement of the chunk-parallel GPU pass with the sequential reference: float symbols over the prefix the single-thread leg covered, int8 soft symbols over
    everything the full-stream reference run produced -- those of the handles the timed steps use (top level; "q8": which instantiation that was) and those of the
    float instantiation the float symbols come from ("float_path")."""
    rs = ref["syms"]
    n = min(len(rs), len(gpu_syms))
    scale = float(np.sqrt(np.mean(np.abs(rs[:n]) ** 2)))
    err = np.abs(gpu_syms[:n] - rs[:n]) / scale
    out = {"symbols_compared": int(n), "frac_within_1e-5": round(float(np.mean(err <= 1e-5)), 6), "frac_bit_identical": round(float(np.mean(err == 0)), 6),
           "median_rel": float(np.median(err)), "p99_rel": float(np.quantile(err, 0.99)), "p99.9_rel": float(np.quantile(err, 0.999)),
           "max_rel": float(err.max())}
    out.update(int8_parity(gpu_soft, ref_soft_full))
    if gpu_soft_float_path is not None:
        out["float_path"] = int8_parity(gpu_soft_float_path, ref_soft_full)
        out["float_path"]["identical_to_the_timed_handles_stream"] = bool(len(gpu_soft) == len(gpu_soft_float_path) and np.array_equal(gpu_soft, gpu_soft_float_path))
    return out


def arm_grid(wl, x_prefix, ref, gpu_syms, gpu_pos):
    """What the symbols beyond 1e-5 ARE (tests/test_demod_gpu.py::test_every_symbol_beyond_tolerance_is_an_arm_flip at bench size): the clock recovery interpolates
    every symbol on one of 128 arms (clock_recovery_mm.cpp:66); the plain-C restatement of the reference (its symbols asserted bit-identical to the compiled
    reference's on this very prefix) and the engine (test tap) both give the interpolation's grid position per symbol. Classes: same position / one grid step
    (1/128 sample: the arm flicker of two trajectories of the timing loop) / anything else."""
    from oracle import pyref
    if not pyref.port_available():
        return None
    ocfg, _, _ = ref_cfgs(wl)
    want, ref_pos = pyref.psk_demod_with_arms(ocfg, x_prefix)
    pinned = bool(np.array_equal(want["syms"].view(np.uint32), ref["syms"].view(np.uint32)))
    n = min(len(ref_pos), len(gpu_pos), len(gpu_syms), len(ref["syms"]))
    rs = ref["syms"][:n]
    scale = float(np.sqrt(np.mean(np.abs(rs) ** 2)))
    err = np.abs(gpu_syms[:n] - rs) / scale
    step = gpu_pos[:n] - ref_pos[:n]
    same, flip = step == 0, np.abs(step) == 1
    other = ~same & ~flip
    amp = np.abs(np.abs(gpu_syms[:n]) - np.abs(rs)) / scale
    ang = np.abs(np.angle(gpu_syms[:n] * np.conj(rs)))
    beyond = err > 1e-5
    nb = max(1, int(beyond.sum()))
    steps = np.abs(step[other]) if other.any() else np.zeros(0, dtype=np.int64)
    hist = {str(k): int((steps == k).sum()) for k in (2, 3, 4)}
    hist[">=5"] = int((steps >= 5).sum())
    return {"symbols": int(n), "restatement_symbols_bit_identical_to_the_reference": pinned,
            "same_arm": {"frac": round(float(same.mean()), 6), "beyond_1e-5": int((err[same] > 1e-5).sum()), "max_rel": float(err[same].max()),
                         "max_amplitude_rel": float(amp[same].max()), "max_angle_rad": float(ang[same].max())},
            "one_arm_step": {"frac": round(float(flip.mean()), 6), "max_rel": float(err[flip].max()) if flip.any() else 0.0},
            "other": int(other.sum()), "other_steps_hist": hist,
            "of_the_symbols_beyond_1e-5": {"count": int(beyond.sum()), "one_arm_step": round(float((beyond & flip).sum()) / nb, 4), "same_arm": round(float((beyond & same).sum()) / nb, 4),
                                           "two_or_more_steps": round(float((beyond & other).sum()) / nb, 4)},
            "what": "where on the clock recovery's 128-arm grid each symbol was interpolated, both sides: symbols beyond 1e-5 are (a) ONE grid step (1/128 sample) from the "
                    "reference's position -- the arm flicker of two trajectories of the timing loop, the bulk --, (b) on the reference's own arm behind a chunk boundary at which "
                    "the carrier loops' hand-off differed (a phase step that decays within the loop's time constant; the larger ones where an order-4 sign detector had just "
                    "disagreed: DESIGN.md 2), (c) a few per million two or more steps off, behind such a boundary. tests/test_demod_gpu.py::"
                    "test_every_symbol_beyond_tolerance_is_an_arm_flip asserts (a) and bounds (b) on streams of 60 - 140 chunks, where (c) does not occur"}


HEADLINE_MAX_BYTES = 4000


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d} if d is not None else None


def headline(out):
    """The compact object bench.py prints as its LAST stdout line: the contract's keys, the roofline and cpu_baseline objects, the parity verdicts as numbers --
    no prose, no per-kernel tables, no other workloads (those are in bench_detail.json). Kept under HEADLINE_MAX_BYTES (tests/test_bench_contract_cpu.py)."""
    h = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    h["config"] = {"workload": cfg.get("workload"), "mode": cfg.get("mode")}
    if out.get("n_gpus", 1) > 1:
        h["config"]["sharding"] = "ONE recording in contiguous chunks per GPU + lock-in overlap, host stitch, no data-path collective"
    h["cadu_per_s"] = out.get("cadu_per_s")
    h["whole_path_GBps"] = out.get("whole_path_GBps")
    h["roofline"] = _pick(out.get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algo_bytes_per_launch", "avg_launch_ms",
                                                "launches_per_step", "kernel_time_frac_of_step"))
    cpu = out.get("cpu_baseline")
    if cpu is not None:
        c = _pick(cpu, ("value", "unit", "cores", "kind", "host_cores"))
        c["sample"] = str(cpu.get("sample", ""))[:160]
        c["volk"] = "generic-order shim (no libvolk on the box)"
        if cpu.get("single_thread"):
            c["single_thread"] = cpu["single_thread"].get("value")
        if cpu.get("all_cores"):
            c["all_cores"] = _pick(cpu["all_cores"], ("value", "cores"))
        h["cpu_baseline"] = c
    else:
        h["cpu_baseline"] = None
    cp = out.get("cadu_parity")
    h["cadu_parity"] = _pick(cp, ("byte_identical", "compared", "reference_cadus", "gpu_cadus_first_pass", "whole_stream", "n_differing", "those_identical_too"))
    sp = out.get("soft_parity")
    if sp is not None:
        s = _pick(sp, ("symbols_compared", "frac_within_1e-5", "max_rel", "int8_compared", "frac_int8_equal", "max_lsb"))
        if sp.get("timed_instantiation"):
            s["timed_instantiation"] = sp["timed_instantiation"]
        if sp.get("float_path"):
            s["float_path"] = _pick(sp["float_path"], ("frac_int8_equal", "max_lsb", "identical_to_the_timed_handles_stream"))
        ag = sp.get("arm_grid")
        if ag:
            s["arm_grid"] = {"symbols": ag.get("symbols"), "other": ag.get("other"), "same_arm_beyond_1e-5": ag["same_arm"].get("beyond_1e-5"),
                             "same_arm_max_angle_rad": ag["same_arm"].get("max_angle_rad"), "one_arm_step_frac": ag["one_arm_step"].get("frac")}
        h["soft_parity"] = s
    else:
        h["soft_parity"] = None
    h["parity_gates"] = out.get("parity_gates")
    h["exact_mode"] = _pick(out.get("exact_mode"), ("value", "samples", "bit_identical_to_the_reference"))
    ck = out.get("check")
    h["check"] = _pick(ck, ("cadus_last_step", "cadus_last_step_all_ranks", "payload_matching_transmitted", "transmitted", "stitched"))
    ks = out.get("kernels") or {}
    h["top_kernels_ms"] = {k: v["ms_per_step"] for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]}
    ow = out.get("other_workloads") or {}
    h["other_workloads"] = {k: {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"),
                                "cadus_identical": (v.get("cadu_parity") or {}).get("byte_identical")} for k, v in ow.items()}
    nr = out.get("next_rows") or {}
    h["next_rows"] = {k: ({"value": v.get("value"), "unit": v.get("unit")} if "error" not in v else {"error": str(v["error"])[:80]}) for k, v in nr.items()}
    h["detail"] = "bench_detail.json"
    return h


def headline_line(out):
    """headline(out) as one JSON line, with the optional parts dropped one by one should it ever exceed HEADLINE_MAX_BYTES."""
    h = headline(out)
    for drop in (None, "next_rows", "other_workloads", "top_kernels_ms", "check", "exact_mode"):
        if drop is not None:
            h.pop(drop, None)
        line = json.dumps(h, separators=(",", ":"))
        if len(line) <= HEADLINE_MAX_BYTES:
            return line
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="metop_ahrpt", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="CADUs per step per GPU (0 = the config's full size)")
    ap.add_argument("--cpu-samples", type=int, default=40_000_000,
                    help="samples of the single-thread / all-cores CPU legs and of the float-symbol comparison (0 = skip every CPU leg)")
    ap.add_argument("--parity-samples", type=int, default=-1,
                    help="samples the reference (its thread-per-block topology) decodes for cadu_parity / the int8 soft parity / cpu_baseline.value "
                         "(-1 = the whole stream, the default: full-stream CADU identity)")
    ap.add_argument("--next-rows", type=int, default=1, help="N=1: also run tools/bench_ndsp.py and tools/bench_dvbs2.py, under next_rows (0 = skip)")
    ap.add_argument("--others", type=int, default=1, help="N=1: also measure the other two single-GPU workloads, compact, under other_workloads (0 = skip)")
    ap.add_argument("--others-parity-samples", type=int, default=400_000_000, help="reference span of the other workloads' parity legs")
    ap.add_argument("--exact-samples", type=int, default=100_000_000,
                    help="samples of the exact_mode leg (exact=1: one sequential lane per loop, bit-identical soft symbols asserted against the reference; 0 = skip)")
    ap.add_argument("--streamed-samples", type=int, default=1 << 30,
                    help="samples of the streamed leg: the host-buffer entry points (sdhip_demod_push / flush / pull), pageable and pinned host memory (0 = skip)")
    ap.add_argument("--chunk-len", type=int, default=0)
    ap.add_argument("--exact", type=int, default=0)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--blocks", type=int, default=0,
                    help="blocks of --frames CADUs the recording consists of (default: one per GPU); --gpus 1 --blocks N decodes on one GPU the very "
                         "recording that --gpus N shards")
    ap.add_argument("--dump", default="", help="write the last step's CADUs of every rank to <prefix>.rank<r>.npy (+ <prefix>.json on rank 0): tests")
    ap.add_argument("--detail", default="", help="where the full result object goes (default: bench_detail.json beside bench.py); stdout carries the compact line only")
    ap.add_argument("--pipeline", action="store_true",
                    help="N=1: overlap the decoder of step i with the demodulator of step i+1 (two host threads, two HIP streams) like the reference's "
                         "thread-per-module pipeline; off by default: the per-kernel HIP-event times of the roofline need the kernels un-overlapped")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # plumbing check on a 1-GPU box only: SDHIP_BENCH_SHARE_GPU=1 puts every rank on device 0 and reduces over gloo
    share_gpu = os.environ.get("SDHIP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    ctx = dict(world=world, rank=rank, local_rank=local_rank, device=device, share_gpu=share_gpu)

    out = run_workload(args, args.workload, args.steps, args.warmup, args.parity_samples, ctx)
    def _bad(o):
        return bool(o and ((o.get("cadu_parity") is not None and not o["cadu_parity"]["byte_identical"]) or (o.get("parity_gates") is not None and not o["parity_gates"]["passed"])
                           or (o.get("exact_mode") is not None and not o["exact_mode"]["bit_identical_to_the_reference"])))
    failed = _bad(out)
    if rank == 0 and world == 1 and args.others and not args.exact and not args.frames and not args.dump:
        # the other two single-GPU workloads, compact: fewer steps, a bounded reference span (GOES' 2 GiB fits it whole)
        others = {}
        for name in WORKLOADS:
            if name == args.workload:
                continue
            torch.cuda.empty_cache()
            import copy
            a2 = copy.copy(args)
            a2.exact_samples = 0    # the exact_mode and streamed legs belong to the driver workload
            a2.streamed_samples = 0
            o = run_workload(a2, name, max(2, min(args.steps, 6)), max(1, min(args.warmup, 2)), args.others_parity_samples, ctx)
            keep = ("value", "unit", "ms_per_step", "steps", "cadu_per_s", "config", "roofline", "soft_parity", "parity_gates", "cadu_parity", "exact_mode", "check", "demod_stats")
            c = {k: o[k] for k in keep if k in o}
            c["config"] = o["config"]["workload"]
            c["cpu_baseline"] = {k: o["cpu_baseline"][k] for k in ("value", "cores", "sample")} if o.get("cpu_baseline") else None
            c["top_kernels_ms"] = {k: v["ms_per_step"] for k, v in sorted(o["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]}
            others[name] = c
            failed = failed or _bad(o)
        out["other_workloads"] = others
    if rank == 0 and world == 1 and args.next_rows and not args.exact and not args.frames and not args.dump:
        # the rows SURVEY.md 8 marks "next", each measured by its own tool (same HIP-event + reference-on-the-host method), compact: the ndsp
        # PSK demodulator chain (f-1) and the DVB-S2 FEC tail (f-2). A failure here is recorded, it never takes the driver line down.
        nxt = {}
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        for key, mod, argv in (("ndsp_psk_demod", "bench_ndsp", ["--steps", "4", "--warmup", "1", "--cpu-samples", "12000000"]),
                               ("dvbs2_fec", "bench_dvbs2", ["--rate", "2/3", "--sigma", "13"]),
                               # (the same decoder on frames that converge after a few update passes: the early exit timed)
                               ("dvbs2_fec_converging", "bench_dvbs2", ["--rate", "2/3", "--front", "0", "--sigma", "10.5", "--sync-frames", "0"]),
                               ("dvbs2_demod_8psk", "bench_dvbs2_demod", []),
                               ("meteor_lrpt_decoder", "bench_lrpt", []),
                               ("fengyun_ahrpt_decoder", "bench_fy3", [])):
            try:
                torch.cuda.empty_cache()
                m = __import__(mod)
                nxt[key] = m.run(m.parse(argv))
            except Exception as e:  # noqa: BLE001
                nxt[key] = {"error": f"{type(e).__name__}: {e}"}
        out["next_rows"] = nxt
    if rank == 0:
        # everything measured goes to a file; stdout carries ONE compact line (the driver keeps only the tail of stdout: round 5's 26.7 KB line did not parse)
        detail = args.detail or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail, "w") as fh:
                json.dump(out, fh, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {detail}: {e}", file=sys.stderr)
        print(headline_line(out), flush=True)
        if failed:
            print("bench.py: parity failed (CADUs differ from the reference's on the same IQ, soft symbols below the workload's floor, or exact mode not bit-identical)", file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_workload(args, workload, n_steps, n_warmup, parity_samples, ctx):
    """One measurement of one workload (see the module docstring); returns the result object on rank 0."""
    import torch
    import torch.distributed as dist
    from satdump_amd import capi, shard, synth

    world, rank, local_rank, device, share_gpu = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["device"], ctx["share_gpu"]
    coll_dev = torch.device("cpu") if share_gpu else device

    wl = WORKLOADS[workload]
    frames = args.frames or wl["frames"]
    frames = max(wl["frames_quantum"], frames // wl["frames_quantum"] * wl["frames_quantum"])
    spec = synth.SynthSpec(**wl["spec"])
    blocks = args.blocks or world
    if blocks % world:
        raise SystemExit("--blocks must be a multiple of the number of GPUs")
    bpr = blocks // world  # blocks per rank
    rec = synth.Recording(spec, frames, blocks=blocks)
    share = rec.samples_per_block * bpr
    dcfg_kw = dict(**wl["demod"], device=local_rank, chunk_len=args.chunk_len, exact=args.exact)
    fcfg_kw = dict(**wl["fec"], device=local_rank)
    overlap = shard.lockin_overlap(wl["demod"], wl["fec"]) if world > 1 else 0
    plan = shard.plan_chunks(rec.n_samples, world, overlap)[rank]
    t_gen = time.perf_counter()
    x = rec.synth_range(plan["read_start"], plan["stop"], device=device)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    n_in = x.numel()
    cadu_bytes = 1024

    soft_cap = 2 * n_in + 64
    d_soft = torch.empty(soft_cap, dtype=torch.int8, device=device)
    cap_frames = frames * bpr + 256
    d_cadu = torch.empty((cap_frames, cadu_bytes), dtype=torch.uint8, device=device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dem = capi.PskDemod(capi.demod_cfg(**dcfg_kw))
    fec = capi.FecDecoder(capi.fec_cfg(**fcfg_kw))

    # ---- parity leg, part 1 (rank 0, N=1): the FIRST pass of fresh handles over the full-size stream, with the float symbols of
    # the prefix the CPU leg will cover. Untimed; it also is the pass that allocates and acquires lock.
    do_cpu = world == 1 and args.cpu_samples > 0 and not args.exact
    parity_gpu = None
    warmup_ms = []
    if do_cpu:
        ncpu = min(n_in, args.cpu_samples)
        sps_in = wl["spec"]["samplerate"] / wl["spec"]["symbolrate"]
        syms_cap = int(ncpu / sps_in * 1.02) + 4096
        d_syms = torch.empty(2 * syms_cap, dtype=torch.float32, device=device)
        # (A) the handles of the timed steps, called the way the timed steps call them (no float symbols asked for: on QPSK the clock recovery's int8 instantiation,
        # k_mm<.., Q8> + k_compact8): its int8 soft symbols and its CADUs are what cadu_parity and soft_parity's int8 figures compare with the reference's
        tw = time.perf_counter()
        ns0 = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap)
        nf0 = fec.process_dev(d_soft.data_ptr(), ns0, d_cadu.data_ptr(), cap_frames)
        torch.cuda.synchronize()
        warmup_ms.append(round((time.perf_counter() - tw) * 1e3, 2))
        q = wl["soft_per_sym"]
        parity_gpu = {"soft": d_soft[:ns0].cpu().numpy(), "cadus": d_cadu[:nf0].cpu().numpy(), "first_pass_stats": dem.stats()}
        # (B) the same first pass through a second fresh handle WITH the float symbols of the prefix (the float instantiation + k_quantize): the float-symbol
        # comparison, and that instantiation's int8 stream beside (A)'s
        dem_f = capi.PskDemod(capi.demod_cfg(**dcfg_kw))
        d_soft_f = torch.empty(soft_cap, dtype=torch.int8, device=device)
        ns_f = dem_f.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft_f.data_ptr(), soft_cap, d_syms.data_ptr(), syms_cap)
        torch.cuda.synchronize()
        parity_gpu["syms"] = d_syms[: 2 * min(syms_cap, ns_f // q)].cpu().numpy().view(np.complex64)
        parity_gpu["soft_float_path"] = d_soft_f[:ns_f].cpu().numpy()
        dem_f.close()
        del d_soft_f
        # the same first pass once more through a second fresh handle with the test tap on (sdhip_demod_set_tap): where on its 128-arm grid the clock recovery
        # interpolated every symbol of the prefix -- soft_parity's arm_grid classification (same trajectory: the engine is deterministic)
        if hasattr(capi.PskDemod, "set_tap"):
            dem_t = capi.PskDemod(capi.demod_cfg(**dcfg_kw))
            dem_t.set_tap(1)
            d_soft_t = torch.empty(soft_cap, dtype=torch.int8, device=device)
            ns_t = dem_t.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft_t.data_ptr(), soft_cap, d_syms.data_ptr(), syms_cap)
            torch.cuda.synchronize()
            if ns_t == ns0:
                parity_gpu["arm_pos"] = d_syms[: 2 * min(syms_cap, ns0 // q)].cpu().numpy().view(np.int64)
            dem_t.close()
            del d_soft_t
        del d_syms

    tot_frames = 0
    tot_soft = 0
    stitched_total = None
    if world == 1:
        def step():
            ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap)
            nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), cap_frames)
            return ns, nf
    else:
        # one cold start per step: fresh handles, their device blocks recycled through the library's pool
        capi.pool_enable(True)
        # boundary frames each rank contributes to the stitch: every frame that can lie in the lock-in overlap, plus a margin
        EDGE = shard.edge_frames(overlap, rec.samples_per_block / frames)
        state = {}
        q_soft = wl["soft_per_sym"]
        sps_nom = wl["spec"]["samplerate"] / wl["spec"]["symbolrate"]
        lock_demod, lock_fec, lock_block = shard.lockin_parts(wl["demod"], wl["fec"])

        def all_gather_np(a):
            t = torch.from_numpy(np.ascontiguousarray(a)).to(coll_dev)
            outv = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outv, t)
            return [o.cpu().numpy() for o in outv]

        def step():
            nonlocal dem, fec
            dem.close()
            fec.close()
            dem = capi.PskDemod(capi.demod_cfg(**dcfg_kw))
            fec = capi.FecDecoder(capi.fec_cfg(**fcfg_kw))
            ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap)
            # where this rank's soft stream continues its predecessor's (a few KB of boundary symbols exchanged), and from there the byte its decoder starts at:
            # on the single stream's Viterbi block grid -- the N lists then stitch into the single stream's list, whole frames (shard.hip)
            lag, turn, agree, before, found = shard.align_ranks(lambda nb: d_soft[ns - nb:ns].cpu().numpy(), lambda nb: d_soft[:nb].cpu().numpy(), ns, q_soft, plan, rank, world,
                                                                all_gather_np)
            start = shard.fec_start(before, lag, q_soft, lock_block, lock_fec, int(lock_demod / sps_nom) * q_soft) if rank and found else 0
            state["align"] = {"lag_symbols": lag, "quarter_turns": turn, "agreement": round(agree, 4), "found": bool(found), "decoder_starts_at_soft_byte": start}
            nf = fec.process_dev(d_soft.data_ptr() + start, ns - start, d_cadu.data_ptr(), cap_frames)
            # stitch on the host from the boundary frames only (the CADU lists themselves stay with their ranks, like the
            # per-module output files of the reference): [count | first EDGE frames | last EDGE frames] per rank
            edge = torch.zeros((2 * EDGE + 1, cadu_bytes), dtype=torch.uint8, device=device)
            edge[0, :8] = torch.tensor(list(int(nf).to_bytes(8, "little")), dtype=torch.uint8, device=device)
            h = min(EDGE, nf)
            edge[1:1 + h] = d_cadu[:h]
            edge[1 + EDGE:1 + EDGE + h] = d_cadu[nf - h:nf]
            edge = edge.to(coll_dev)
            allv = [torch.empty_like(edge) for _ in range(world)]
            dist.all_gather(allv, edge)
            if rank == 0:
                hv = [a.cpu().numpy() for a in allv]
                counts = [int.from_bytes(bytes(a[0, :8]), "little") for a in hv]
                heads = [a[1:1 + min(EDGE, c)] for a, c in zip(hv, counts)]
                tails = [a[1 + EDGE:1 + EDGE + min(EDGE, c)] for a, c in zip(hv, counts)]
                state["drops"] = shard.stitch_plan(heads, tails, counts, EDGE, whole_frames=True)
                state["counts"] = counts
            return ns, nf

    for _ in range(n_warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        warmup_ms.append(round((time.perf_counter() - tw) * 1e3, 2))
    barrier()
    capi.prof_reset()
    capi.prof_enable(True)
    pipelined = args.pipeline and n_steps > 1 and world == 1
    t0 = time.perf_counter()
    if not pipelined:
        for _ in range(n_steps):
            ns, nf = step()
            tot_soft += ns
            tot_frames += nf
    else:
        # The reference runs psk_demod and the decoder as two modules on two threads joined by a FIFO (pipeline_run.cpp:72-104):
        # while the decoder works on buffer i the demodulator already fills buffer i+1. Same here: two host threads (ctypes
        # releases the GIL), each engine on its own HIP stream, two .soft buffers. All K demodulator passes and all K decoder
        # passes lie inside the timed region.
        import threading
        bufs = [d_soft, torch.empty(soft_cap, dtype=torch.int8, device=device)]
        res = {}

        def run_dem(i):
            res["ns", i] = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, bufs[i % 2].data_ptr(), soft_cap)

        def run_fec(i):
            res["nf", i] = fec.process_dev(bufs[i % 2].data_ptr(), res["ns", i], d_cadu.data_ptr(), cap_frames)

        run_dem(0)
        for i in range(n_steps):
            th = None
            if i + 1 < n_steps:
                th = threading.Thread(target=run_dem, args=(i + 1,))
                th.start()
            run_fec(i)
            if th is not None:
                th.join()
        for i in range(n_steps):
            tot_soft += res["ns", i]
            tot_frames += res["nf", i]
        ns, nf = res["ns", n_steps - 1], res["nf", n_steps - 1]
    barrier()
    dt = time.perf_counter() - t0
    capi.prof_enable(False)
    prof = capi.prof_get()
    last_nf = nf

    # samples of the recording each rank OWNS (the overlap is re-processed work, not throughput)
    own = float((plan["stop"] - plan["own_start"]) * n_steps)
    dt_all, samples_all, frames_all = shard.reduce_metrics(dt, own, float(tot_frames), device=None if share_gpu else device)

    # ---- correctness of what was timed: every CADU of the last step must be one of the transmitted frames of this rank's
    # range (its block plus, with N>1, the overlap frames in front of it); N>1: the stitched list has no duplicates and no gaps
    check = None
    if not args.no_check:
        got = d_cadu[:last_nf].cpu().numpy()
        want, want_payload = set(), set()
        order = {}
        for b in range(rank * bpr, (rank + 1) * bpr):
            plain = rec.plain_cadus(b)
            want |= {bytes(p) for p in plain}
            want_payload |= {bytes(p[4:]) for p in plain}
            for i, p in enumerate(plain):
                order[bytes(p[4:20])] = (b - rank * bpr) * frames + i
        if args.dump:
            np.save(f"{args.dump}.rank{rank}.npy", got)
        if world > 1:
            lead = -(-overlap // (rec.samples_per_block // frames)) + 2
            pn = np.resize(synth._PN, 255 * spec.rs_i)
            for p in rec.frames(rank * bpr * frames - lead, rank * bpr * frames):
                p = p.copy()
                if spec.derand:
                    p[4:] ^= pn
                want.add(bytes(p))
                want_payload.add(bytes(p[4:]))
        ok = sum(1 for g in got if bytes(g) in want)
        bad_at = [i for i, g in enumerate(got) if bytes(g[4:]) not in want_payload]
        # where the output leaves the transmitted order: (position in the output, id before, id after); the stream is periodic, so
        # id + 1 modulo the frame count is "no gap"
        ids = [order.get(bytes(g[4:20]), -1) for g in got]
        nfr = frames * bpr
        gaps = [(i, ids[i - 1], ids[i]) for i in range(1, len(ids)) if ids[i] >= 0 and ids[i - 1] >= 0 and ids[i] != (ids[i - 1] + 1) % nfr]
        ok_payload = len(got) - len(bad_at)  # the 4-byte ASM is not RS protected: channel errors stay in it
        check = {"cadus_last_step": int(last_nf), "cadus_matching_transmitted": int(ok), "payload_matching_transmitted": int(ok_payload),
                 "transmitted": int(frames * bpr), "not_matching_at": bad_at[:6] + (["..."] + bad_at[-3:] if len(bad_at) > 9 else bad_at[6:9]),
                 "first_id": ids[0] if ids else None, "order_breaks": gaps[:8]}
        if world > 1:
            c = torch.tensor([float(last_nf), float(ok), float(ok_payload)], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(c)
            if rank == 0:
                drops, counts = state["drops"], state["counts"]
                stitched_total = int(sum(counts) - sum(drops))
                check = {"cadus_last_step_all_ranks": int(c[0].item()), "cadus_matching_transmitted": int(c[1].item()),
                         "payload_matching_transmitted": int(c[2].item()), "transmitted": int(frames * blocks),
                         "stitched": stitched_total, "dropped_as_decoded_twice": [int(d) for d in drops], "rank0_alignment": state.get("align")}
                if args.dump:
                    with open(args.dump + ".json", "w") as fh:
                        json.dump({"drops": [int(d) for d in drops], "counts": [int(c) for c in counts]}, fh)

    if rank == 0:
        dst = dem.stats()
        fst = fec.stats()
        steps = n_steps
        passes = max(1, n_warmup + steps + (1 if do_cpu else 0)) if world == 1 else 1
        nsym = dst.symbols_out // passes
        n_rs = n_in if not dst.resample_interp else (n_in * dst.resample_interp) // dst.resample_decim
        nsoft = tot_soft // steps
        algo = algorithmic_bytes(wl, n_in, n_rs, nsym, nsoft, (tot_frames // steps) * cadu_bytes, 8, q8="k_compact8" in prof)
        kernels = {k: {"ms_per_step": round(v[0] / steps, 4), "launches_per_step": round(v[1] / steps, 2)} for k, v in prof.items()}
        for k, v in kernels.items():
            if k in algo and v["ms_per_step"] > 0:
                v["algo_GBps"] = round(algo[k] / (v["ms_per_step"] * 1e-3) / 1e9, 2)
                # measured HBM traffic over algorithmic bytes, per step (PMC passes of these very sources, else absent)
                tr, _src = pmc_traffic(workload, k)
                if tr and algo[k] > 0:
                    v["traffic_over_algorithmic"] = round(tr * v["launches_per_step"] / algo[k], 3)
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else None
        roof = None
        if dom is not None:
            ms_step = prof[dom][0] / steps
            launches = prof[dom][1] / steps
            bytes_step = algo.get(dom, 0.0)
            achieved = bytes_step / (ms_step * 1e-3) / 1e9 if ms_step > 0 else 0.0
            traffic, traffic_src = pmc_traffic(workload, dom)
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "algo_bytes_per_launch": round(bytes_step / max(launches, 1e-9)), "avg_launch_ms": round(ms_step / max(launches, 1e-9), 4),
                    "launches_per_step": round(launches, 2), "kernel_time_frac_of_step": round(ms_step / (dt / steps * 1e3), 4)}
        cpu = None
        sparity = None
        cparity = None
        if do_cpu:
            npar = n_in if parity_samples < 0 else min(n_in, max(parity_samples, ncpu))
            xh = x[:npar].cpu().numpy().view(np.complex64)
            cpu, ref, full = cpu_baseline(wl, xh, ncpu)
            del xh
            sparity = soft_parity(parity_gpu["syms"], parity_gpu["soft"], parity_gpu.get("soft_float_path"), ref, full["soft"])
            sparity["timed_instantiation"] = "k_mm<Q8> + k_compact8" if "k_compact8" in prof else "k_mm (float rows) + k_quantize"
            if "arm_pos" in parity_gpu:
                sparity["arm_grid"] = arm_grid(wl, x[:ncpu].cpu().numpy().view(np.complex64), ref, parity_gpu["syms"], parity_gpu["arm_pos"])
            fp = parity_gpu["first_pass_stats"]
            sparity["what"] = (f"first pass of fresh handles over the full {n_in}-sample stream (chunk-parallel mode) against the sequential reference: int8 soft symbols "
                               f"of the timed steps' own handles (no float symbols requested) over its first {full['samples']} samples; float symbols (a second fresh "
                               f"handle, the float instantiation) over its first {ncpu} samples")
            sparity["first_pass_chunks"] = {"chunks": fp.chunks, "re_run": fp.chunks_fixed, "accepted_by_tolerance": fp.chunks_inexact, "let_through": fp.chunks_forced}
            ref_cadus, gpu_cadus = full["cadu"], parity_gpu["cadus"]
            m = min(len(ref_cadus), len(gpu_cadus))
            neq = np.flatnonzero((ref_cadus[:m] != gpu_cadus[:m]).any(axis=1)) if m else np.zeros(0, dtype=np.int64)
            # frames of the compared span that are NOT one of the transmitted frames (RS could not correct them and rs_usecheck is off, or a
            # miscorrection): identity must hold on those too -- their bytes depend on the soft symbols
            tx = set()
            for b in range(blocks):
                tx |= {bytes(pl) for pl in rec.plain_cadus(b)}
            off_tx = [i for i in range(m) if bytes(ref_cadus[i]) not in tx]
            cparity = {"reference_cadus": int(len(ref_cadus)), "gpu_cadus_first_pass": int(len(gpu_cadus)), "compared": int(m),
                       "reference_span_samples": int(full["samples"]), "whole_stream": bool(full["samples"] == n_in),
                       "byte_identical": bool(m > 0 and len(neq) == 0), "differing_frames": [int(v) for v in neq[:8]], "n_differing": int(len(neq)),
                       "frames_not_matching_transmitted_in_span": len(off_tx), "those_at": off_tx[:8],
                       "those_identical_too": bool(all(np.array_equal(ref_cadus[i], gpu_cadus[i]) for i in off_tx)),
                       "tail": "frames the reference still had inside its block hand-offs at EOF are dropped by its stop() (module_demod_base.cpp); the GPU "
                               "pass flushes nothing either: both lists end within a frame or two of the end of the stream"}
        exact_leg = None
        streamed = None
        gates = None
        if do_cpu:
            # ---- gates: what the chunk-parallel mode promises for this workload, enforced -- a bench line outside them is a failed line
            gates = parity_gates(workload, sparity)
            # ---- exact mode (exact=1: every loop one sequential lane, the reference's float operations in its order): the mode that IS bit-identical,
            # timed on a prefix and checked bit for bit against the reference's soft stream
            if args.exact_samples > 0:
                ne = min(n_in, args.exact_samples, full["samples"])
                de = capi.PskDemod(capi.demod_cfg(**dict(dcfg_kw, exact=1)))
                d_se = torch.empty(2 * ne + 64, dtype=torch.int8, device=device)
                torch.cuda.synchronize()
                te = time.perf_counter()
                nse = de.process_dev(x.data_ptr(), ne, capi.FMT_CF32, d_se.data_ptr(), 2 * ne + 64)
                torch.cuda.synchronize()
                te = time.perf_counter() - te
                se = d_se[:nse].cpu().numpy()
                k = min(len(se), len(full["soft"]))
                same = bool(k > 0.99 * nse and np.array_equal(se[:k], full["soft"][:k]))
                exact_leg = {"value": round(ne / te / 1e6, 3), "unit": "Msamples/s", "samples": int(ne), "seconds": round(te, 2), "soft_bytes_compared": int(k),
                             "bit_identical_to_the_reference": same, "what": "psk_demod with exact=1 (one sequential lane per loop stage), IQ resident in HBM, demodulator only"}
                de.close()
                del d_se, de
            # ---- the path a SatDump module calls: host buffers through sdhip_demod_push / flush / pull (PCIe inclusive; never `value`)
            if args.streamed_samples > 0:
                nst = min(n_in, args.streamed_samples)
                xs = x[:nst].cpu().numpy()
                streamed = {"samples": int(nst), "unit": "Msamples/s",
                            "what": "cf32 samples in host memory -> sdhip_demod_push in 4 Mi-sample calls, sdhip_demod_pull after every call (what plugin/sdhip_plugin.cpp's file loop "
                                    "does), flush + pull at the end -> int8 soft symbols in host memory. Inside: staging copy by the library's copy threads into two pinned "
                                    "buffers, H2D + kernels + D2H on a worker thread. Two passes over the samples on ONE handle (one stream); the second is quoted (the first "
                                    "also allocates the pinned buffers)"}
                piece = 4 << 20
                sink = np.empty(64 << 20, dtype=np.int8)
                for kind in ("pageable", "pinned"):
                    src = xs if kind == "pageable" else torch.from_numpy(xs).pin_memory().numpy()
                    ds = capi.PskDemod(capi.demod_cfg(**dcfg_kw))
                    times, nsoft = [], 0
                    for _rep in range(2):
                        ts = time.perf_counter()
                        for a0 in range(0, nst, piece):
                            ds.push(src[a0:a0 + piece])
                            while True:
                                g_ = ds.pull(out=sink)
                                nsoft += len(g_)
                                if len(g_) < sink.size:
                                    break
                        ds.flush()
                        while True:
                            g_ = ds.pull(out=sink)
                            nsoft += len(g_)
                            if len(g_) < sink.size:
                                break
                        times.append(time.perf_counter() - ts)
                    ds.close()
                    streamed[kind] = round(nst / times[1] / 1e6, 1)
                    streamed[kind + "_GB_per_s"] = round(nst * 8 / times[1] / 1e9, 2)
                    streamed[kind + "_first_pass"] = round(nst / times[0] / 1e6, 1)
                    streamed["soft_bytes_both_passes"] = int(nsoft)
                    del src
                del xs
        q = wl["soft_per_sym"]
        sps_in = wl["spec"]["samplerate"] / wl["spec"]["symbolrate"]
        algo_per_sample = 8 + 2 * q / sps_in + (q * wl["conv_rate"] / 8.0) / sps_in
        if world == 1:
            sharding = "one continuous stream on one GPU"
        else:
            sharding = (f"ONE recording of {rec.n_samples} samples cut into {world} contiguous chunks, one per GPU, each read from {overlap} samples early "
                        f"(lock-in overlap), cold start per step, every rank's decoder started on the single stream's Viterbi block grid (found from a few KB of boundary symbols: "
                        f"sdhip_shard_align), per-rank CADU lists stitched on the host from the boundary frames compared whole; no data-path collective")
        out = {
            "metric": "Msamples/s IQ through PSK demod -> Viterbi -> RS (HBM-resident cf32)",
            "value": round(samples_all / dt_all / 1e6, 3), "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": n_warmup,
            "ms_per_step": round(dt_all / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{workload} (BASELINE {wl['baseline']}): {wl['spec']['constellation'].upper()} {wl['spec']['symbolrate']:.0f} sym/s @ "
                                   f"{wl['spec']['samplerate'] / 1e6:g} Msps cf32, conv {wl['spec']['conv']}, RS(255,223) I=4, {frames * bpr} CADUs = "
                                   f"{share} samples ({share * 8 / 1e9:.3f} GB) per GPU per step",
                       "mode": "exact" if args.exact else "chunk-parallel (speculate + certify)", "sharding": sharding,
                       "module_overlap": "decoder of step i overlaps the demodulator of step i+1 (two host threads, two HIP streams)" if pipelined
                       else "none (modules back to back)"},
            "cadu_per_s": round((stitched_total * steps if stitched_total is not None else frames_all) / dt_all, 1),
            "algo_bytes_per_sample": round(algo_per_sample, 3),
            "whole_path_GBps": round(samples_all * algo_per_sample / dt_all / 1e9, 3),
            "roofline": roof, "cpu_baseline": cpu, "soft_parity": sparity, "parity_gates": gates, "cadu_parity": cparity, "exact_mode": exact_leg, "streamed": streamed,
            "check": check,
            "demod_stats": {"chunks": dst.chunks, "chunks_fixed": dst.chunks_fixed, "chunks_rotated": dst.chunks_rotated,
                            "chunks_inexact": dst.chunks_inexact, "chunks_forced": dst.chunks_forced, "freq_hz": round(dst.freq_hz, 2)},
            "fec_stats": {"vit_respec": fst.vit_respec, "tb_respec": fst.tb_respec, "viterbi_ber": round(fst.viterbi_ber, 4),
                          "blocks": fst.blocks, "frames_out": fst.frames_out},
            "kernels": kernels, "input_gen_s": round(t_gen, 2), "warmup_ms": warmup_ms,
        }
        dem.close()
        fec.close()
        del x, d_soft, d_cadu
        return out
    dem.close()
    fec.close()
    return None


if __name__ == "__main__":
    main()
