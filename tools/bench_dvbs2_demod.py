#!/usr/bin/env python3
"""BASELINE.json configs[4] on one MI355X: DVB-S2 8PSK (MODCOD 13 = rate 2/3, normal FECFRAMEs, roll-off 0.2) at 45 Msym/s, two samples per symbol,
BASEBAND samples resident in HBM -> BBFRAMEs, through the module-shaped handle (sdhip_dvbs2_demod_*: front end, PL synchroniser, frame-parallel
PLL, soft demapper stage, LDPC in the reference build's 16-frame groups, BCH, BB descrambler). One step = one pass over `--frames` PLFRAMEs of
a periodic recording (so that consecutive steps continue ONE stream: loop states, the synchroniser's ring and the decoder groups carry over).
Reported: Msym/s, Msamples/s, frames/s; per-kernel HIP-event times; the dominant kernel against its HBM roofline; every BBFRAME of the first
timed step checked against the transmitted ones; the reference's blocks and classes chained the way the module chains them on a bounded sample
of the same samples -- its BBFRAMEs must be ours (parity) and its rate on the host is the cpu_baseline (one thread: the chain run block after
block).   usage: tools/bench_dvbs2_demod.py [--frames 512] [--steps 3] [--esn0 10] [--cpu-frames 24]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODCOD, SYMRATE, SPS, ALPHA, LOOP_BW = 13, 45e6, 2, 0.2, 0.002


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048, help="PLFRAMEs per step (a multiple of --base)")
    ap.add_argument("--base", type=int, default=32, help="distinct BBFRAMEs of the periodic recording")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--esn0", type=float, default=10.0)
    ap.add_argument("--trials", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="frames per LDPC decode call of the reference build being replaced (SSE4.1: 16)")
    ap.add_argument("--freq-prop", type=float, default=0.01, help="the module's freq_prop_factor (its default, module_dvbs2_demod.cpp:32-33; the reference chain beside it runs "
                                                                  "without the feedback: the module's is thread-timed)")
    ap.add_argument("--cfo-rad", type=float, default=1e-4, help="carrier offset of the recording, rad per SAMPLE (1e-4 at 90 Msps = 1.43 kHz; the reference's frame PLL pulls "
                                                                "that in within a dozen frames, 2e-4 within a hundred: measured on its compiled blocks)")
    ap.add_argument("--cpu-frames", type=int, default=576, help="frames' worth of samples the reference chain decodes on the host, acquisition stretch included (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1, help="all-cores leg: that many processes run the reference chain at once (-1 = one per core up to 128, 0 = skip)")
    ap.add_argument("--exact", type=int, default=0)
    return ap.parse_args(argv)


def run(args) -> dict:
    import torch
    torch.zeros(1, device="cuda")
    from satdump_amd import capi, dvbs2, synth, synth_dvbs2 as sd
    c = sd.modcod_cfg(MODCOD, 0)
    raw = 90 + c["slots"] * 90
    base = args.base
    nfr = max(base, args.frames // base * base)
    bb = sd.bbframes_random(0, c["rate"], base, seed=5)
    t0 = time.perf_counter()
    syms = sd.plframes(MODCOD, 0, bb).reshape(-1)
    t_gen = time.perf_counter() - t0
    spec = synth.SynthSpec(constellation="qpsk", samplerate=SYMRATE * SPS, symbolrate=SYMRATE, rrc_alpha=ALPHA, amplitude=0.5, cfo_hz=0.0, esn0_db=args.esn0, seed=5,
                           timing_offset=0.3)
    d_clean, _ = synth.modulate_torch(syms, dataclass_noiseless(spec), torch.device("cuda"), periodic=True)
    nb = d_clean.numel()
    reps = nfr // base
    g = torch.Generator(device="cuda").manual_seed(11)
    sigma = float(np.sqrt(SPS / (2.0 * 10 ** (args.esn0 / 10)))) * spec.amplitude
    d_x = torch.view_as_real(d_clean).repeat(reps, 1).contiguous()
    d_x += sigma * torch.randn(d_x.shape, device="cuda", generator=g)
    n = nb * reps
    del d_clean
    # the carrier offset, over the whole recording (a whole number of turns per step, so that consecutive steps continue ONE stream)
    turns = round(args.cfo_rad * n / (2.0 * np.pi))
    w_cfo = 2.0 * np.pi * turns / n
    if turns:
        for a in range(0, n, 1 << 24):
            b = min(n, a + (1 << 24))
            ph = torch.remainder(torch.arange(a, b, device="cuda", dtype=torch.float64) * w_cfo, 2.0 * np.pi).to(torch.float32)
            cs, sn = torch.cos(ph), torch.sin(ph)
            re, im = d_x[a:b, 0].clone(), d_x[a:b, 1]
            d_x[a:b, 0] = re * cs - im * sn
            d_x[a:b, 1] = re * sn + im * cs
        del ph, cs, sn, re
    # the demapper table: data the module builds on the host with the reference's constellation_t. On the GPU box the compiled reference class is the
    # prebuilt checker library; the bench only takes the TABLE from it (what the plugin takes from libsatdump_core)
    from oracle import pyref
    fref = pyref.S2FrontRef()
    lut_b, lut_p = fref.lut(MODCOD, 0), pyref.s2_lut_phase_ref(MODCOD, 0)
    params = {"samplerate": SYMRATE * SPS, "symbolrate": SYMRATE, "rrc_alpha": ALPHA, "pll_bw": LOOP_BW, "modcod": MODCOD, "freq_prop_factor": args.freq_prop,
              "ldpc_trials": args.trials}
    dem = dvbs2.DVBS2Demod(params, lut_b, lut_p, mem=dvbs2.TorchMem("cuda"), exact=bool(args.exact), batch=args.batch)
    fb = dem.bbframe_bytes
    cap = nfr + 64
    d_out = torch.zeros(cap * fb, dtype=torch.uint8, device="cuda")
    sent = {bytes(r): i for i, r in enumerate(bb)}
    firsts = []
    call_ms = []  # wall time of the untimed calls: the FIRST call of a stream is its cold start (narrow M&M windows, the PLL's serial frames, acquisition)
    for _ in range(args.warmup):
        torch.cuda.synchronize()
        tc = time.perf_counter()
        k = dem.process_dev(d_x.data_ptr(), n, capi.FMT_CF32, d_out.data_ptr(), cap)
        torch.cuda.synchronize()
        call_ms.append(round((time.perf_counter() - tc) * 1e3, 3))
        firsts.append(d_out[:k * fb].cpu().numpy().reshape(k, fb).copy())
    st0 = dem.stats
    torch.cuda.synchronize()
    capi.prof_reset()
    capi.prof_enable(True)
    outs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        k = dem.process_dev(d_x.data_ptr(), n, capi.FMT_CF32, d_out.data_ptr(), cap)
        outs.append(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    capi.prof_enable(False)
    prof = capi.prof_get()
    st = dem.stats
    got = d_out[:outs[-1] * fb].cpu().numpy().reshape(outs[-1], fb)
    hits = [sent.get(bytes(r), -1) for r in got]
    ok = [h for h in hits if h >= 0]
    in_order = all((b - a) % base == 1 for a, b in zip(ok[:-1], ok[1:])) and len(ok) == len(hits)
    kern = {k2: round(v[0] / args.steps, 3) for k2, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    nsym = nfr * raw
    out = {"metric": "DVB-S2 8PSK baseband -> BBFRAMEs: Msym/s of the configured symbol rate's stream through the whole demodulator module, baseband resident in HBM",
           "value": round(nsym / dt / 1e6, 1), "unit": "Msym/s", "Msamples_per_s": round(n / dt / 1e6, 1), "frames_per_s": round(sum(outs) / args.steps / dt, 1),
           "realtime_factor_at_45_Msym_per_s": round(nsym / dt / SYMRATE, 2), "ms_per_step": round(dt * 1e3, 3), "steps": args.steps, "dtype": "f32 + int8",
           "config": {"workload": f"BASELINE configs[4]: MODCOD {MODCOD} (8PSK 2/3), normal FECFRAMEs, roll-off {ALPHA}, {SPS} samples per symbol ({SYMRATE * SPS / 1e6:.0f} Msps for "
                                  f"{SYMRATE / 1e6:.0f} Msym/s), Es/N0 {args.esn0} dB, {nfr} PLFRAMEs = {n} cf32 samples ({n * 8 / 1e6:.0f} MB) per step, periodic recording of {base} "
                                  f"distinct BBFRAMEs with fresh noise on every repetition, carrier offset {w_cfo:.3e} rad/sample ({w_cfo * SYMRATE * SPS / (2 * np.pi) / 1e3:.2f} kHz), freq_prop_factor "
                                  f"{args.freq_prop}, max {args.trials} LDPC trials in groups of {args.batch}, "
                                  + ("serial schedules (exact)" if args.exact else "chunk-parallel front end + frame-parallel PLL")},
           "bbframes_per_step": outs, "all_bbframes_are_transmitted_ones_in_order": bool(in_order), "frames_not_matching": len(hits) - len(ok),
           "stats": {k2: (round(v, 6) if isinstance(v, float) else v) for k2, v in st.items() if k2 not in ("frames", "pls", "freq")},
           "pll_schedule_per_step": {"lanes": (st["pll_lanes"] - st0["pll_lanes"]) // args.steps, "rerun": (st["pll_rerun"] - st0["pll_rerun"]) / args.steps,
                                     "forced": st["pll_forced"] - st0["pll_forced"], "serial_frames_at_stream_start": st0["pll_serial_frames"]},
           "kernels_ms": kern, "synthesis_s": round(t_gen, 1),
           "cold_start": {"untimed_calls_ms": call_ms, "note": "wall time of the calls in front of the timed ones, same input each: the first is a new stream's (allocation, acquisition, the "
                                                               "default M&M windows, the PLL's first 65 536 symbols on the serial lane), the later ones the steady state's"}}
    # ---- roofline of the dominant kernel. Algorithmic bytes per step: front-end lane stages 8 B in + 8 B out per sample (k_afc* is not used here: the
    # DVB-S2 front end has no Costas stage: k_chunks<AgcFir> + k_mm); k_ldpc_trial: 2 x the check-to-bit messages + 2 x the LLRs per frame and update pass
    dom = next(iter(kern)) if kern else None
    ldpc = capi.LdpcDecoder(framesize=0, rate="2/3", batch=args.batch)
    upd = float(st["ldpc_trials"])
    algo = {"k_ldpc_trial": sum(outs) / args.steps * max(upd, 1.0) * (2 * ldpc.info.msg_bytes_per_frame + 2 * 64800),
            "k_mm": n * 8 + nsym * 8, "k_chunks_AgcFirStage": n * 16, "k_s2_pll_lanes": nsym * 16, "k_s2_demap": nsym * 11, "k_s2_plsync_search": nsym * 8}
    if dom:
        key = next((k2 for k2 in algo if dom.startswith(k2)), None)
        if key and kern[dom] > 0:
            ach = algo[key] / (kern[dom] * 1e-3) / 1e9
            import bench as _b
            lps = prof[dom][1] / args.steps if dom in prof else 1.0
            tr_b, tr_1, tr_src = _b.pmc_traffic_per_step("dvbs2", dom, lps)  # per STEP, like algorithmic_bytes_per_step
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": tr_b,
                               "traffic_per_launch": tr_1, "launches_per_step": lps, "traffic_source": tr_src,
                               "algorithmic_bytes_per_step": int(algo[key]), "note": "ldpc update passes per frame taken from the last group's trial count" if key == "k_ldpc_trial" else ""}
    whole = (n * 8 + sum(outs) / args.steps * fb) / dt / 1e9
    out["whole_path"] = {"algorithmic_GB_per_s": round(whole, 1), "frac_of_hbm_peak": round(whole / 8000.0, 4), "note": "8 B per baseband sample in + the BBFRAME bytes out"}
    # ---- the reference beside it: its blocks and classes chained the way the module chains them, on the first cpu-frames frames' worth of samples
    if args.cpu_frames > 0:
        m = min(n, args.cpu_frames * raw * SPS)
        xs = torch.view_as_complex(d_x[:m]).cpu().numpy()
        import dvbs2_cpu_chain as cc
        want, nxr, stage, tr, fbatch = cc.chain(xs, MODCOD, SYMRATE, SPS, ALPHA, LOOP_BW, args.trials, args.batch)
        whits = [sent.get(bytes(r), -1) for r in want]
        first = firsts[0] if firsts else got
        # the reference's k-th frame is the stream's k-th PLFRAME (both PL synchronisers emit the same frames; tests/test_dvbs2_gpu.py): compare position by
        # position where the reference's frame is a transmitted one -- its cold-started loop needs some frames to settle, the lanes' header estimates do not
        fhits = [sent.get(bytes(r), -1) for r in first[:len(want)]]
        common = [k for k in range(min(len(want), len(first))) if whits[k] >= 0]
        same = all(np.array_equal(want[k], first[k]) for k in common)
        out["cpu_baseline"] = {"value": round(nxr / stage["total"] / 1e6, 3), "unit": "Msym/s", "cores": 1, "kind": "reference",
                               "sample": f"the first {m} samples ({args.cpu_frames} frames' worth): AGC, RRC filter, M&M, S2PLSyncBlock, S2PLLBlock, S2BBToSoft, BBFrameLDPC (SIMD width {fbatch}), "
                                         "BBFrameBCH, BB descrambler, one after the other on one thread",
                               "stage_seconds": {k: v for k, v in stage.items() if k != "total"}, "ldpc_trials": [int(v) for v in tr[:8]]}
        if args.cpu_procs != 0:
            # the same chain in P independent processes at once (an instance per core: what the host does flat out), on the same sample
            import subprocess
            import tempfile
            P = args.cpu_procs if args.cpu_procs > 0 else max(1, min(os.cpu_count() or 1, 128))
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "x.npy")
                m_all = min(m, 64 * raw * SPS)  # (the all-cores leg keeps its 64-frame sample: P copies of it are in flight)
                np.save(f, xs[:m_all])
                cmd = [sys.executable, os.path.join(ROOT, "tools", "dvbs2_cpu_chain.py"), f, str(MODCOD), str(SYMRATE), str(SPS), str(ALPHA), str(LOOP_BW), str(args.trials), str(args.batch)]
                tw = time.perf_counter()
                procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
                outs_p = [p.communicate()[0] for p in procs]
                wall = time.perf_counter() - tw
            good = [o.split() for o in outs_p if o.strip()]
            if good:
                out["cpu_baseline"]["all_cores"] = {"value": round(sum(float(g[1]) for g in good) / wall / 1e6, 2), "unit": "Msym/s", "cores": len(good), "host_cores": os.cpu_count(),
                                                    "sample": f"{len(good)} independent processes x the same {m_all} samples in {wall:.1f} s wall (process start-up included)",
                                                    "slowest_chain_s": round(max(float(g[0]) for g in good), 2)}
        first_ref = next((k for k, h in enumerate(whits) if h >= 0), None)
        first_our = next((k for k, h in enumerate(fhits) if h >= 0), None)
        out["parity_sample"] = {"reference_frames": int(len(want)), "reference_frames_that_are_transmitted_ones": int(sum(h >= 0 for h in whits)),
                                "our_frames_that_are_transmitted_ones_on_the_same_positions": int(sum(h >= 0 for h in fhits)),
                                "first_transmitted_frame": {"reference": first_ref, "ours": first_our},
                                "reference_hits_head": whits[:24], "our_hits_head": fhits[:24],
                                "frames_compared": len(common), "byte_identical": bool(same and len(common) >= 1),
                                "acquisition": "the reference's cold-started decision-directed loop pulls the offset in over its first frames (its blocks compiled in place lock a "
                                               "dozen frames in at 1e-4 rad/sample, ~100 at 2e-4); ours walks the first 65 536 symbols with that same loop, then anchors every lane on "
                                               "the frame headers (two consecutive headers give the frequency, the neighbouring branches are tried when the chain disagrees) and "
                                               "delivers frames the reference is still searching for: compared are the positions at which the reference's frame is a transmitted one"}
    return out


def dataclass_noiseless(spec):
    """the same SynthSpec with the noise switched off (it is added per repetition on the device instead)"""
    import dataclasses
    return dataclasses.replace(spec, esn0_db=200.0)


def main():
    print(json.dumps(run(parse())), flush=True)


if __name__ == "__main__":
    main()
