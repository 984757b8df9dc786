#!/usr/bin/env python3
"""The ndsp PSK demodulator chain (satdump::ndsp::PSKDemodHierBlock's RRC -> AGC -> M&M -> Costas) on one MI355X, samples
resident in HBM: complex samples/s of sdhip_ndsp_psk_demod_work_dev in the chunk-parallel mode, per-kernel HIP-event times, symbol
parity of the first call against the REFERENCE hier block on a prefix (its four member blocks on their own threads: oracle/_ref), and
that reference's own rate on the host.   usage: tools/bench_ndsp.py [--samples 1073741824] [--constellation qpsk] [--steps 5]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1 << 30)
    ap.add_argument("--constellation", default="qpsk")
    ap.add_argument("--samplerate", type=float, default=6e6)
    ap.add_argument("--symbolrate", type=float, default=2e6)  # the block's defaults (psk_demod.h:35-36)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-samples", type=int, default=24_000_000)
    return ap.parse_args(argv)


def run(args) -> dict:
    import torch
    torch.zeros(1, device="cuda")
    from oracle import pyref
    from satdump_amd import capi, synth

    # one periodic block (symbols wrap, the offset is a whole number of cycles per block), tiled in HBM
    nsym_blk = 1 << 20
    rng = np.random.default_rng(11)
    if args.constellation == "bpsk":
        a = (rng.integers(0, 2, nsym_blk) * 2.0 - 1.0).astype(np.complex128)
    else:
        a = ((rng.integers(0, 2, nsym_blk) * 2.0 - 1.0) + 1j * (rng.integers(0, 2, nsym_blk) * 2.0 - 1.0)) / np.sqrt(2.0)
    spec = synth.SynthSpec(constellation=args.constellation, samplerate=args.samplerate, symbolrate=args.symbolrate, rrc_alpha=0.35, amplitude=0.4, cfo_hz=9000.0,
                           esn0_db=12.0, seed=11)
    blk, cfo = synth.modulate(a, spec, periodic=True)
    reps = max(1, args.samples // len(blk))
    n = reps * len(blk)
    d_x = torch.from_numpy(blk.view(np.float32)).cuda().repeat(reps)
    d_y = torch.zeros(2 * (n // 2 + 64), dtype=torch.float32, device="cuda")

    c = capi.NdspPskCfg()
    L = capi.lib()
    L.sdhip_ndsp_psk_cfg_default(C.byref(c))
    c.constellation = capi.BPSK if args.constellation == "bpsk" else capi.QPSK
    c.samplerate, c.symbolrate = args.samplerate, args.symbolrate
    h = L.sdhip_ndsp_psk_demod_create(C.byref(c))
    assert h, capi.last_error()

    def step():
        ns = L.sdhip_ndsp_psk_demod_work_dev(h, C.c_void_p(d_x.data_ptr()), n, C.c_void_p(d_y.data_ptr()), n // 2 + 64)
        assert ns > 0, capi.last_error()
        return ns

    # first call from the cold state: what the reference computes from sample 0 -- parity on a prefix
    ns0 = step()
    st = capi.DemodStats()
    L.sdhip_ndsp_psk_demod_get_stats(h, C.byref(st))
    first = dict(chunks=st.chunks, re_run=st.chunks_fixed, accepted_by_tolerance=st.chunks_inexact, let_through=st.chunks_forced)
    ncpu = min(n, args.cpu_samples)
    xs = np.tile(blk, -(-ncpu // len(blk)))[:ncpu]
    nref = pyref.NdspRef()
    t0 = time.time()
    want = nref.run("psk_demod_cc", {"constellation": args.constellation, "samplerate": args.samplerate, "symbolrate": args.symbolrate}, xs)
    t_cpu = time.time() - t0
    got = d_y[: 2 * len(want)].cpu().numpy().view(np.complex64)
    lock = 60000
    err = np.abs(got[lock:] - want[lock:]) / np.sqrt(np.mean(np.abs(want[lock:]) ** 2))
    if args.constellation == "qpsk":
        hard = (np.sign(got.real) != np.sign(want.real)) | (np.sign(got.imag) != np.sign(want.imag))
    else:
        hard = np.sign(got.real) != np.sign(want.real)
    parity = dict(symbols_compared=int(len(want) - lock), median_rel=float(np.median(err)), frac_within_1e5=float(np.mean(err <= 1e-5)),
                  frac_beyond_1e3=float(np.mean(err > 1e-3)), max_rel=float(err.max()), hard_decisions_differing=int(hard[lock:].sum()),
                  note="steady state: from symbol 60000 on (the reference loop is still pulling the 9 kHz offset in before that)")

    for _ in range(args.warmup):
        step()
    capi.prof_enable(True)
    capi.prof_reset()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        ns = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    prof = capi.prof_get()
    capi.prof_enable(False)
    kern = {k: round(v[0] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}
    L.sdhip_ndsp_psk_demod_get_stats(h, C.byref(st))
    res = {
        "metric": "ndsp psk_demod_cc complex samples/s, samples resident in HBM", "value": round(n / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
        "config": {"workload": f"{args.constellation} {args.samplerate / 1e6:g} Msps / {args.symbolrate / 1e6:g} Msym/s, Es/N0 12 dB, {cfo:.0f} Hz offset, {n} samples per call, "
                               "block defaults (rrc 0.35 / 31 taps, agc 1e-4 / 0.6, M&M 8.7e-3, loop_bw 0.004)"},
        "symbols_per_call": int(ns), "steady_chunks": {"chunks": st.chunks, "re_run": st.chunks_fixed}, "first_call_chunks": first, "kernels_ms_per_step": kern,
        "parity_vs_reference_first_call": parity, "pll_freq_hz": round(float(st.freq_hz), 1),
        "cpu_reference": {"value": round(ncpu / t_cpu / 1e6, 2), "unit": "Msamples/s", "threads": 6, "kind": "reference",
                          "sample": f"first {ncpu} samples, the hier block's own topology: rrc, agc, rec, pll, splitter, snr estimator threads + feeder"}}
    # per-kernel algorithmic bytes per step: a lane stage reads and writes every sample once (8 + 8 B), the clock recovery reads the samples and writes the symbols
    steps_b = {"k_chunks": 16 * n, "k_fir_window": 16 * n, "k_fir": 16 * n, "k_mm": 8 * n + 8 * int(ns), "k_quantize": 16 * int(ns), "k_derotate": 16 * int(ns)}
    res["roofline"] = _roofline("ndsp", kern, steps_b, "8 B in + 8 B out per sample of the dominant stage", {k: v[1] / args.steps for k, v in prof.items()})
    res["whole_path"] = {"algorithmic_GB_per_s": round((8 * n + 8 * int(ns)) / dt / 1e9, 1), "frac_of_hbm_peak": round((8 * n + 8 * int(ns)) / dt / 1e9 / 8000.0, 5)}
    L.sdhip_ndsp_psk_demod_destroy(h)
    del d_x, d_y
    return res



def _roofline(tag, kern, algo_bytes, note, launches=None):
    """the dominant kernel of the line against the HBM roof (bench.py's object): algorithmic bytes of that kernel per step / its HIP-event time per step; traffic
    from the PMC profile of this very bench when one was committed for these kernel sources (bench.pmc_traffic)"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as _b
    dom = max(kern, key=kern.get) if kern else None
    if not dom or kern[dom] <= 0:
        return None
    key = next((k for k in algo_bytes if dom.startswith(k)), None)
    if key is None:
        return {"bound": "hbm", "kernel": dom, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None, "note": "no byte model for this kernel: " + note}
    ach = algo_bytes[key] / (kern[dom] * 1e-3) / 1e9
    lps = float((launches or {}).get(dom, 1.0))
    tr, tr1, src = _b.pmc_traffic_per_step(tag, dom, lps)  # per STEP, like algo_bytes_per_step (x launches per step)
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5), "traffic": tr, "traffic_per_launch": tr1,
            "launches_per_step": lps, "traffic_source": src,
            "algo_bytes_per_step": int(algo_bytes[key]), "ms_per_step": kern[dom], "note": note}


def main():
    print(json.dumps(run(parse())), flush=True)


if __name__ == "__main__":
    main()
