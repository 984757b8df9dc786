#!/usr/bin/env python3
"""A/B of engine switches on ONE workload in ONE process (the input is synthesised once, the reference decodes the prefix once):
for every configuration -- a set of SDHIP_* environment switches, read by the engines when the handles are created / at every call --
fresh handles, warm-up passes, timed passes, per-kernel HIP-event times, soft-symbol parity of the first pass against the reference on
the first --cpu-samples samples, CADU count.  usage: tools/ab_demod.py --workload metop_ahrpt "A=1,B=2" "C=3" ...  ("" = defaults)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="metop_ahrpt")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-samples", type=int, default=40_000_000)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--pool", type=int, default=0, help="1: the handles of consecutive configurations recycle their device blocks (sdhip_pool_enable) instead of freeing and allocating")
    ap.add_argument("configs", nargs="*", default=[""])
    args = ap.parse_args()
    import torch
    from oracle import pyref
    from satdump_amd import capi, synth
    wl = bench.WORKLOADS[args.workload]
    if args.pool:
        capi.pool_enable(True)
    frames = args.frames or wl["frames"]
    frames = max(wl["frames_quantum"], frames // wl["frames_quantum"] * wl["frames_quantum"])
    dev = torch.device("cuda", 0)
    rec = synth.Recording(synth.SynthSpec(**wl["spec"]), frames, blocks=1)
    x = rec.synth_range(0, rec.n_samples, device=dev)
    n_in = x.numel()
    ncpu = min(n_in, args.cpu_samples)
    ref = None
    if ncpu > 0:
        ref, ref_cadus, _, _ = bench.ref_decode(pyref.best(), wl, x[:ncpu].cpu().numpy(), want_syms=True)
    soft_cap = 2 * n_in + 64
    d_soft = torch.empty(soft_cap, dtype=torch.int8, device=dev)
    d_cadu = torch.empty((frames + 256, 1024), dtype=torch.uint8, device=dev)
    sps = wl["spec"]["samplerate"] / wl["spec"]["symbolrate"]
    syms_cap = int(ncpu / sps * 1.02) + 4096
    d_syms = torch.empty(2 * syms_cap, dtype=torch.float32, device=dev)
    base_env = dict(os.environ)
    for cfg in args.configs:
        os.environ.clear()
        os.environ.update(base_env)
        for kv in [c for c in cfg.split(",") if c]:
            k, v = kv.split("=")
            os.environ[k] = v
        dem = capi.PskDemod(capi.demod_cfg(**wl["demod"]))
        fec = capi.FecDecoder(capi.fec_cfg(**wl["fec"]))
        ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap, d_syms.data_ptr(), syms_cap)
        nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), frames + 256)
        torch.cuda.synchronize()
        st0 = dem.stats()
        par = None
        if ref is not None:
            q = wl["soft_per_sym"]
            gs = d_syms[: 2 * min(syms_cap, ns // q)].cpu().numpy().view(np.complex64)
            gq = d_soft[: min(ns, syms_cap * q)].cpu().numpy()
            par = bench.soft_parity(gs, gq, ref, ref["soft"])
            m = min(len(ref_cadus), nf)
            par["cadus_identical"] = bool(np.array_equal(ref_cadus[:m], d_cadu[:m].cpu().numpy()))
            par = {k: par[k] for k in ("frac_within_1e-5", "p99.9_rel", "max_rel", "frac_int8_equal", "max_lsb", "cadus_identical")}
        for _ in range(args.warmup):
            ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap)
            nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), frames + 256)
        torch.cuda.synchronize()
        capi.prof_reset()
        capi.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap)
            nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), frames + 256)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        capi.prof_enable(False)
        prof = capi.prof_get()
        top = {k.replace("k_chunks<", "").replace("Stage>", ""): round(v[0] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:14]}
        st = dem.stats()
        print(json.dumps({"cfg": cfg or "(defaults)", "ms_per_step": round(dt * 1e3, 3), "cadus": int(nf), "first_pass": {"fixed": st0.chunks_fixed, "inexact": st0.chunks_inexact, "forced": st0.chunks_forced},
                          "steady": {"chunks": st.chunks, "fixed": st.chunks_fixed}, "parity": par, "kernels_ms": top}), flush=True)
        dem.close()
        fec.close()


if __name__ == "__main__":
    main()
