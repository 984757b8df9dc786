#!/usr/bin/env python3
"""bench.py -- Msamples/s of IQ through psk_demod -> Viterbi -> deframe -> derand -> RS on MI355X.

One "step" = one pass of the whole hot path (sdhip_demod_process_dev + sdhip_fec_process_dev, include/sdhip.h) over
one batch of synthetic IQ that is ALREADY RESIDENT IN HBM when the timed region starts; soft symbols and CADUs
stay in HBM too. The workload is BASELINE.json configs[1] (GOES HRIT: BPSK 927 ksym/s @ 3 Msps cf32, r=1/2
Viterbi + RS(255,223) I=4, ~2 GB) unless --workload selects another config. N>1 (torchrun, one rank per GPU):
every rank demodulates/decodes its own independent baseband stream of the same size -- the path shards
stream-parallel with no data-path collective -- so scaling is "weak" and value = samples of all
ranks / max-over-ranks time.

Prints ONE JSON line on rank 0. Extra objects: "roofline" (dominant kernel, HIP-event timed on the launch stream
via sdhip_prof_*), "cpu_baseline" (the compiled reference oracle/_ref -- or the restatement -- on a bounded sample of
the same workload, rank 0 / N=1 only), "cadu_per_s", "kernels" (per-kernel ms/step).
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
        frames_quantum=309, frames=16 * 309, q=1, soft_per_sym=1, conv_rate=0.5),
    # BASELINE.json configs[2]: MetOp AHRPT QPSK + punctured r=3/4 + RS, 16 GB cf32 @ 6 Msps
    "metop_ahrpt": dict(
        spec=dict(constellation="qpsk", samplerate=6e6, symbolrate=2333333, conv="3/4-metop", nrzm=False, esn0_db=10.0, amplitude=0.25,
                  cfo_hz=3000.0, seed=3),
        demod=dict(samplerate=6e6, symbolrate=2333333, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.003),
        fec=dict(decoder=1, viterbi_ber_thresold=0.28, viterbi_outsync_after=10),
        frames_quantum=21, frames=21 * 7280, q=2, soft_per_sym=2, conv_rate=0.75),
    # BASELINE.json configs[3] per-GPU share: JPSS HRD QPSK 15 Msym/s @ 30 Msps, 16 GB cf32 per GPU
    "npp_hrd": dict(
        spec=dict(constellation="qpsk", samplerate=30e6, symbolrate=15e6, conv="1/2", nrzm=True, esn0_db=8.0, amplitude=0.4, cfo_hz=20000.0,
                  seed=4),
        demod=dict(samplerate=30e6, symbolrate=15e6, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.002),
        fec=dict(constellation="qpsk", cadu_size=8192, viterbi_ber_thresold=0.3, viterbi_outsync_after=20, derandomize=1, nrzm=1, rs_i=4,
                 rs_type=1, rs_usecheck=1),
        frames_quantum=1, frames=131072, q=2, soft_per_sym=2, conv_rate=0.5),
}


def make_input(wl, device, seed_offset, frames):
    """Synthesise the rank's baseband stream in HBM (periodic: symbol sequence and carrier wrap seamlessly, so
    consecutive steps look like one continuous stream to the stateful engines)."""
    from satdump_amd import synth
    spec = synth.SynthSpec(**wl["spec"])
    spec.seed += seed_offset
    seed = spec.seed
    while True:
        cadus = synth.make_cadus(frames, seed=seed)
        bits = np.unpackbits(cadus.reshape(-1))
        if not spec.nrzm or int(bits.sum()) % 2 == 0:  # NRZ-M level must wrap too
            break
        seed += 1000
    plain = synth.make_cadus(frames, seed=seed, derand=False)
    syms = synth.frames_to_symbols(cadus, spec, circular=True)
    x, cfo = synth.modulate_torch(syms, spec, device, periodic=True)
    return x, plain, spec


def algorithmic_bytes(wl, n_in, n_rs, nsym, nsoft, nblk_bytes, cadu_bytes_out, in_bytes_per_sample):
    """Per-step minimal HBM traffic of each kernel when stages are NOT fused: read + write once."""
    q = wl["soft_per_sym"]
    return {
        "k_convert": n_in * (in_bytes_per_sample + 8),
        "k_resample": n_in * 8 + n_rs * 8,
        "k_resample_byoffset": n_in * 8 + n_rs * 8,
        "k_resample_period": n_in * 8 + n_rs * 8,
        "k_chunks<AgcStage>": n_rs * 16,
        "k_fir": n_rs * 16,
        "k_fir_window": n_rs * 16,
        "k_chunks<CostasStage>": n_rs * 16,
        "k_mm": n_rs * 8 + nsym * 8,
        "k_quantize": nsym * (8 + q),
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
    produced by tools/gpu_round.sh + tools/pmc_summary.py with the guide's gfx950 x2 correction on FETCH_SIZE). PMC passes cannot
    run inside this process, so the figure is the one of the most recent committed profile of this workload; None if there is none."""
    import csv
    import glob
    short = {"goes_hrit": "goes", "metop_ahrpt": "metop", "npp_hrd": "npp"}[workload]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{short}_pmc.csv")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        rows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
    hdr, rows = rows[0], rows[1:]
    for r in rows:
        if r and r[0] == kernel:
            return float(r[hdr.index("traffic_bytes_per_dispatch")]), os.path.basename(files[-1])
    return None, os.path.basename(files[-1])


def cpu_baseline(wl, x_host, max_seconds_hint=20.0):
    """The reference's own code (oracle/_ref, else the restatement) on a bounded sample of the same stream, one host thread."""
    from oracle import pyref
    kind = "reference" if pyref.ref_available() else "port"
    orc = pyref.best()
    d = wl["demod"]
    cons = {"bpsk": pyref.BPSK, "qpsk": pyref.QPSK}[d["constellation"]]
    ocfg = pyref.demod_cfg(samplerate=d["samplerate"], symbolrate=d["symbolrate"], constellation=cons, rrc_alpha=d["rrc_alpha"], pll_bw=d["pll_bw"],
                           max_sps=d.get("max_sps", 4.0))
    t0 = time.perf_counter()
    r = orc.psk_demod(ocfg, x_host, want_syms=False)
    t1 = time.perf_counter()
    f = wl["fec"]
    if f.get("decoder", 0) == 1:
        out = orc.metop_decode(r["soft"], ber_thr=f["viterbi_ber_thresold"], outsync_after=f["viterbi_outsync_after"])
    else:
        ofec = pyref.fec_cfg(constellation=cons, nrzm=f["nrzm"], rs_usecheck=f["rs_usecheck"], viterbi_ber_thresold=f["viterbi_ber_thresold"],
                             viterbi_outsync_after=f["viterbi_outsync_after"])
        out = orc.concat_decode(ofec, r["soft"])
    t2 = time.perf_counter()
    n = len(x_host)
    return {
        "value": round(n / (t2 - t0) / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": kind,
        "sample": f"first {n} samples of rank 0's stream, demod {t1 - t0:.2f}s + FEC {t2 - t1:.2f}s, {len(out['cadu'])} CADUs, single thread "
                  f"(generic-order VOLK shim, gcc -O2)",
        "cadus": int(len(out["cadu"])),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="goes_hrit", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="CADUs per step per GPU (0 = the config's full size)")
    ap.add_argument("--cpu-samples", type=int, default=40_000_000, help="samples of the CPU-baseline leg (0 = skip)")
    ap.add_argument("--chunk-len", type=int, default=0)
    ap.add_argument("--exact", type=int, default=0)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap the decoder of step i with the demodulator of step i+1 (two host threads, two HIP streams) like the reference's "
                         "thread-per-module pipeline; off by default: the per-kernel HIP-event times of the roofline need the kernels un-overlapped")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from satdump_amd import capi

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

    wl = WORKLOADS[args.workload]
    frames = args.frames or wl["frames"]
    frames = max(wl["frames_quantum"], frames // wl["frames_quantum"] * wl["frames_quantum"])
    t_gen = time.perf_counter()
    x, plain, spec = make_input(wl, device, seed_offset=100 * rank, frames=frames)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    n_in = x.numel()

    dem = capi.PskDemod(capi.demod_cfg(**wl["demod"], device=local_rank, chunk_len=args.chunk_len, exact=args.exact))
    fec = capi.FecDecoder(capi.fec_cfg(**wl["fec"], device=local_rank))
    soft_cap = 2 * n_in + 64
    d_soft = torch.empty(soft_cap, dtype=torch.int8, device=device)
    d_soft2 = torch.empty(soft_cap, dtype=torch.int8, device=device)  # second .soft buffer of the two-stage pipeline
    cap_frames = frames + 64
    d_cadu = torch.empty((cap_frames, 1024), dtype=torch.uint8, device=device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), soft_cap)
        nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), cap_frames)
        return ns, nf

    warmup_ms = []
    for _ in range(args.warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        warmup_ms.append(round((time.perf_counter() - tw) * 1e3, 2))  # the first call also allocates and acquires lock (untimed)
    barrier()
    capi.prof_reset()
    capi.prof_enable(True)
    tot_frames = 0
    tot_soft = 0
    pipelined = args.pipeline and args.steps > 1
    t0 = time.perf_counter()
    if not pipelined:
        for _ in range(args.steps):
            ns, nf = step()
            tot_soft += ns
            tot_frames += nf
    else:
        # The reference runs psk_demod and the decoder as two modules on two threads joined by a FIFO (pipeline_run.cpp:72-104):
        # while the decoder works on buffer i the demodulator already fills buffer i+1. Same here: two host threads (ctypes
        # releases the GIL), each engine on its own HIP stream, two .soft buffers. All K demodulator passes and all K decoder
        # passes lie inside the timed region.
        import threading
        bufs = [d_soft, d_soft2]
        res = {}

        def run_dem(i):
            res["ns", i] = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, bufs[i % 2].data_ptr(), soft_cap)

        def run_fec(i):
            res["nf", i] = fec.process_dev(bufs[i % 2].data_ptr(), res["ns", i], d_cadu.data_ptr(), cap_frames)

        run_dem(0)
        for i in range(args.steps):
            th = None
            if i + 1 < args.steps:
                th = threading.Thread(target=run_dem, args=(i + 1,))
                th.start()
            run_fec(i)
            if th is not None:
                th.join()
        for i in range(args.steps):
            tot_soft += res["ns", i]
            tot_frames += res["nf", i]
        ns, nf = res["ns", args.steps - 1], res["nf", args.steps - 1]
    barrier()
    dt = time.perf_counter() - t0
    capi.prof_enable(False)
    prof = capi.prof_get()
    last_nf = nf

    from satdump_amd import shard
    dt_all, samples_all, frames_all = shard.reduce_metrics(dt, float(n_in * args.steps), float(tot_frames), device=None if share_gpu else device)

    # ---- correctness of what was timed: every CADU of the last step must be one of the transmitted frames
    check = None
    if not args.no_check:
        got = d_cadu[:last_nf].cpu().numpy()
        want = {bytes(p) for p in plain}
        want_payload = {bytes(p[4:]) for p in plain}
        ok = sum(1 for g in got if bytes(g) in want)
        ok_payload = sum(1 for g in got if bytes(g[4:]) in want_payload)  # the 4-byte ASM is not RS protected: channel errors stay in it
        check = {"cadus_last_step": int(last_nf), "cadus_matching_transmitted": int(ok), "payload_matching_transmitted": int(ok_payload),
                 "transmitted": int(frames)}

    if rank == 0:
        dst = dem.stats()
        fst = fec.stats()
        steps = args.steps
        nsym = dst.symbols_out // max(1, (args.warmup + steps))
        n_rs = n_in if not dst.resample_interp else (n_in * dst.resample_interp) // dst.resample_decim
        nsoft = tot_soft // steps
        algo = algorithmic_bytes(wl, n_in, n_rs, nsym, nsoft, 0, (tot_frames // steps) * 1024, 8)
        kernels = {k: {"ms_per_step": round(v[0] / steps, 4), "launches_per_step": round(v[1] / steps, 2)} for k, v in prof.items()}
        for k, v in kernels.items():
            if k in algo and v["ms_per_step"] > 0:
                v["algo_GBps"] = round(algo[k] / (v["ms_per_step"] * 1e-3) / 1e9, 2)
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else None
        roof = None
        if dom is not None:
            ms_step = prof[dom][0] / steps
            launches = prof[dom][1] / steps
            bytes_step = algo.get(dom, 0.0)
            achieved = bytes_step / (ms_step * 1e-3) / 1e9 if ms_step > 0 else 0.0
            traffic, traffic_src = pmc_traffic(args.workload, dom)
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "algo_bytes_per_launch": round(bytes_step / max(launches, 1e-9)), "avg_launch_ms": round(ms_step / max(launches, 1e-9), 4),
                    "launches_per_step": round(launches, 2), "kernel_time_frac_of_step": round(ms_step / (dt / steps * 1e3), 4)}
        cpu = None
        if world == 1 and args.cpu_samples > 0:
            ncpu = min(n_in, args.cpu_samples)
            xh = x[:ncpu].cpu().numpy()
            cpu = cpu_baseline(wl, xh)
        q = wl["soft_per_sym"]
        sps_in = wl["spec"]["samplerate"] / wl["spec"]["symbolrate"]
        algo_per_sample = 8 + 2 * q / sps_in + (q * wl["conv_rate"] / 8.0) / sps_in
        out = {
            "metric": "Msamples/s IQ through PSK demod -> Viterbi -> RS (HBM-resident cf32)",
            "value": round(samples_all / dt_all / 1e6, 3), "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(dt_all / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['spec']['constellation'].upper()} {wl['spec']['symbolrate']:.0f} sym/s @ "
                                   f"{wl['spec']['samplerate'] / 1e6:g} Msps cf32, conv {wl['spec']['conv']}, RS(255,223) I=4, {frames} CADUs = "
                                   f"{n_in} samples ({n_in * 8 / 1e9:.3f} GB) per GPU per step",
                       "mode": "exact" if args.exact else "chunk-speculative", "sharding": f"{world} independent stream(s), one per GPU",
                       "module_overlap": "decoder of step i overlaps the demodulator of step i+1 (two host threads, two HIP streams)" if pipelined
                       else "none (modules back to back)"},
            "cadu_per_s": round(frames_all / dt_all, 1),
            "algo_bytes_per_sample": round(algo_per_sample, 3),
            "whole_path_GBps": round(samples_all * algo_per_sample / dt_all / 1e9, 3),
            "roofline": roof, "cpu_baseline": cpu, "check": check,
            "demod_stats": {"chunks": dst.chunks, "chunks_fixed": dst.chunks_fixed, "chunks_rotated": dst.chunks_rotated,
                            "chunks_inexact": dst.chunks_inexact, "chunks_forced": dst.chunks_forced, "freq_hz": round(dst.freq_hz, 2)},
            "fec_stats": {"vit_respec": fst.vit_respec, "tb_respec": fst.tb_respec, "viterbi_ber": round(fst.viterbi_ber, 4),
                          "blocks": fst.blocks, "frames_out": fst.frames_out},
            "kernels": kernels, "input_gen_s": round(t_gen, 2), "warmup_ms": warmup_ms,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
