#!/usr/bin/env python3
"""meteor_lrpt_decoder (the Viterbi27-based plugin decoders) on one MI355X, soft symbols resident in HBM: frames/s of
sdhip_lrpt_process_dev (correlator walk -> rotate_soft -> Viterbi27 -> NRZ-M / derand -> RS x 4), per-kernel HIP-event times, the CADUs of a prefix
against the reference module's loop on the reference's own classes (oracle/_ref), and that loop's own rate on the host.
usage: tools/bench_lrpt.py [--frames 8192] [--steps 4] [--cpu-frames 256]"""
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
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-frames", type=int, default=256)
    ap.add_argument("--sigma", type=float, default=20.0)
    return ap.parse_args(argv)


def run(args) -> dict:
    import torch
    torch.zeros(1, device="cuda")
    from oracle import pyref
    from satdump_amd import capi
    from tests.test_lrpt_gpu import lrpt_soft

    base = 256  # distinct frames, tiled (the decoder does not care that frames repeat)
    soft, _ = lrpt_soft(base, seed=21, sigma=args.sigma, lead=1234)
    lead, body = soft[:1234], soft[1234:]
    reps = max(1, args.frames // base)
    d_soft = torch.cat([torch.from_numpy(lead).cuda(), torch.from_numpy(body).cuda().repeat(reps)])
    n = int(d_soft.numel())
    nfr = reps * base
    d_out = torch.zeros((nfr + 8) * 1024, dtype=torch.uint8, device="cuda")
    L = capi.lib()
    cfg = capi.LrptCfg()
    L.sdhip_lrpt_cfg_default(C.byref(cfg))

    def one():
        h = L.sdhip_lrpt_create(C.byref(cfg))
        assert h, capi.last_error()
        k = L.sdhip_lrpt_process_dev(h, C.c_void_p(d_soft.data_ptr()), n, C.c_void_p(d_out.data_ptr()), nfr + 8)
        assert k >= 0, capi.last_error()
        L.sdhip_lrpt_destroy(h)
        return k

    L.sdhip_pool_enable(1)  # a handle per step: its buffers come back from the pool instead of hipMalloc
    for _ in range(args.warmup):
        one()
    capi.prof_enable(True)
    capi.prof_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks = [one() for _ in range(args.steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = capi.prof_get()
    kern = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}
    capi.prof_enable(False)
    out = {"metric": "frames_per_s", "value": round(ks[-1] / dt, 1), "unit": "CADU/s", "ms_per_step": round(dt * 1e3, 3), "steps": args.steps,
           "config": {"workload": f"meteor_lrpt_decoder, {nfr} frames of 16384 soft bytes (r=1/2 k=7 QPSK, 1024-byte CADUs, RS(255,223) x 4), sigma {args.sigma} on +-70"},
           "soft_MB_per_s": round(n / dt / 1e6, 1), "frames_out": int(ks[-1]), "kernels_ms": dict(sorted(kern.items(), key=lambda kv: -kv[1])[:8]), "dtype": "u8"}
    # algorithmic bytes per step: the Viterbi stage reads the frames' soft bytes once and writes the decoded bits (the gather in front reads / writes them once more),
    # the correlator reads the hard bits (n / 8), RS reads and writes the frames
    steps_b = {"k_vit2_acs": n + nfr * 1024, "k_vit2_tb": nfr * 1024, "k_vit2_prep": 2 * n, "k_lrpt_gather": 2 * n, "k_lrpt_hard": n + n / 8, "k_lrpt_spec": n / 8, "k_lrpt_chain": n / 8,
               "k_rs": 2 * nfr * 1020, "k_vit_ber": n / 8 + nfr * 1024}
    out["roofline"] = _roofline("lrpt", kern, steps_b, "soft bytes in + decoded bytes out of the dominant kernel", {k: v[1] / args.steps for k, v in prof.items()})
    out["whole_path"] = {"algorithmic_GB_per_s": round((n + ks[-1] * 1024) / dt / 1e9, 2), "frac_of_hbm_peak": round((n + ks[-1] * 1024) / dt / 1e9 / 8000.0, 5)}
    if args.cpu_frames > 0 and pyref.ref_available():
        m = 1234 + args.cpu_frames * 16384
        s = d_soft[:m].cpu().numpy()
        t1 = time.perf_counter()
        want = pyref.ref().lrpt_decode(s, False)["cadu"]
        t2 = time.perf_counter()
        got = d_out[: len(want) * 1024].cpu().numpy().reshape(-1, 1024)
        k = min(len(want), args.cpu_frames - 1)
        out["cpu_baseline"] = {"value": round(args.cpu_frames / (t2 - t1), 1), "unit": "CADU/s", "cores": 1, "kind": "reference",
                               "sample": f"the first {args.cpu_frames} frames: the module's loop (Correlator, rotate_soft, Viterbi27, derand_ccsds, ReedSolomon) on one thread"}
        out["parity_sample"] = {"frames_compared": int(k), "byte_identical": bool(np.array_equal(got[:k], want[:k]))}
    return out



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
