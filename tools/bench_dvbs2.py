#!/usr/bin/env python3
"""DVB-S2 FEC tail on one MI355X (BASELINE.json configs[4]'s decoder: LDPC soft decode + hard-decision repack + BCH), frames resident in
HBM: frames/s and coded Mbit/s, trial launches, k_ldpc_trial's HIP-event time against its HBM roofline (algorithmic bytes per frame and
update pass = 2 x the check-to-bit message state + 2 x the frame's LLRs), and the reference's decoder (oracle/_ref, one thread, its
16-frames-per-call SSE4.1 build when present) on a sample.  usage: tools/bench_dvbs2.py [--rate 2/3] [--frames 4096] [--sigma 13]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--framesize", type=int, default=0)
    ap.add_argument("--rate", default="2/3")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--sigma", type=float, default=13.0)
    ap.add_argument("--trials", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-frames", type=int, default=64)
    ap.add_argument("--front", type=int, default=1, help="1: the frames enter as QPSK PLFRAMEs through sdhip_s2_bb_to_soft_dev (needs oracle/_ref for the demapper table)")
    ap.add_argument("--sync-frames", type=int, default=128, help="frames the PL synchroniser and the frame PLL are timed on (0 = skip)")
    ap.add_argument("--esn0", type=float, default=8.0, help="Es/N0 of the PLFRAMEs, dB")
    return ap.parse_args(argv)


def run(args) -> dict:
    import torch
    torch.zeros(1, device="cuda")
    from oracle import pyref
    from satdump_amd import capi
    from tests import dvbs2_util
    rc = capi.S2_RATES[args.rate]
    ldpc = capi.LdpcDecoder(framesize=args.framesize, rate=args.rate, batch=args.batch)
    bch = capi.BchDecoder(framesize=args.framesize, rate=args.rate)
    n, k = ldpc.info.code_len, ldpc.info.data_len
    nf = args.frames // args.batch * args.batch
    rng = np.random.default_rng(5)
    base = 256  # distinct code words; the noise is per frame
    bb = np.zeros((base, k // 8), dtype=np.uint8)
    bb[:, :bch.kbch // 8] = rng.integers(0, 256, (base, bch.kbch // 8), dtype=np.uint8)
    ref = pyref.Dvbs2Ref(sse=pyref.Dvbs2Ref.available(True) and args.batch == 16)
    bb = ref.bch_encode(args.framesize, rc, bb)
    cw = dvbs2_util.encode(args.framesize, rc, np.unpackbits(bb, axis=1))
    tx = torch.from_numpy(np.where(cw > 0, -20.0, 20.0).astype(np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    soft = torch.clamp(torch.round(tx[torch.arange(nf, device="cuda") % base] + args.sigma * torch.randn((nf, n), device="cuda", generator=g)), -127, 127).to(torch.int8)
    work = torch.empty_like(soft)
    d_tr = torch.zeros(nf // args.batch, dtype=torch.int32, device="cuda")
    d_pack = torch.zeros((nf, k // 8), dtype=torch.uint8, device="cuda")
    d_corr = torch.zeros(nf, dtype=torch.int32, device="cuda")

    # --front 1 (default where the compiled reference is there to hand over its demapper table): the frames enter as PL-synchronised QPSK
    # PLFRAMEs -- header with the PLS code word, the code word's bit pairs on I / Q (bit 0 -> +), PL-scrambled, complex noise -- and
    # sdhip_s2_bb_to_soft_dev (PLS decode, descrambling, table demapping, de-interleaver) produces the decoder's input in every step
    front = None
    import ctypes as C
    QPSK_MODCOD = {"1/4": 1, "1/3": 2, "2/5": 3, "1/2": 4, "3/5": 5, "2/3": 6, "3/4": 7, "4/5": 8, "5/6": 9, "8/9": 10, "9/10": 11}
    if args.front and pyref.S2FrontRef.available() and args.rate in QPSK_MODCOD:
        modcod = QPSK_MODCOD[args.rate]
        fref = pyref.S2FrontRef()
        lut = fref.lut(modcod, args.framesize)
        nsym = n // 2
        x, y, z = 1, 0x3ffff, np.zeros(2 * 131072, dtype=np.uint8)
        for i in range(2 * 131072):  # Gold sequence n = 0 (ETSI EN 302 307-1 5.5.4)
            z[i] = (x ^ y) & 1
            x = ((((x >> 7) ^ x) & 1) << 18 | x) >> 1
            y = ((((y >> 10) ^ (y >> 7) ^ (y >> 5) ^ y) & 1) << 18 | y) >> 1
        rn = (z[:nsym] | (z[131072:131072 + nsym] << 1)).astype(np.int64)
        a = 2.0 / 3.0 / np.sqrt(2.0)  # the table's nominal point: sample * const_amp (3) = (sqrt 2, sqrt 2)
        sym = ((1.0 - 2.0 * cw[:, 0::2]) + 1j * (1.0 - 2.0 * cw[:, 1::2])) * a * np.exp(1j * np.pi / 2 * rn)[None, :]
        stride = 90 + nsym + 6
        base_fr = np.zeros((base, stride), dtype=np.complex64)
        base_fr[:, 90:90 + nsym] = sym
        pcw = int(dvbs2_util.pls_codewords()[(modcod << 2) | (args.framesize << 1)])
        base_fr[:, 26:90] = np.exp(1j * np.pi / 4) * np.where(np.array([(pcw >> (63 - q)) & 1 for q in range(64)]) == 1, -1.0, 1.0)
        sig = a * np.sqrt(2.0) * 10 ** (-args.esn0 / 20.0) / np.sqrt(2.0)  # per component, Es = 2 a^2
        d_base = torch.from_numpy(base_fr.view(np.float32)).cuda()
        d_fr = d_base[torch.arange(nf, device="cuda") % base] + sig * torch.randn((nf, 2 * stride), device="cuda", generator=g)
        d_pls = torch.zeros(nf, dtype=torch.int32, device="cuda")
        front = dict(modcod=modcod, stride=stride, nsym=nsym)

    def step():
        if front:
            r = capi.lib().sdhip_s2_bb_to_soft_dev(0, front["modcod"], args.framesize, 0, C.c_void_p(d_fr.data_ptr()), front["stride"], nf, lut.ctypes.data_as(C.c_void_p), 256,
                                                   C.c_void_p(work.data_ptr()), C.c_void_p(d_pls.data_ptr()))
            assert r == n, capi.last_error()
        else:
            work.copy_(soft)
        launches = ldpc.decode_dev(work.data_ptr(), nf, args.trials, d_tr.data_ptr())
        bch.pack_dev(work.data_ptr(), n, nf, d_pack.data_ptr(), k // 8)
        bch.decode_dev(d_pack.data_ptr(), nf, k // 8, d_corr.data_ptr())
        return launches

    step()
    torch.cuda.synchronize()
    capi.prof_reset()
    capi.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launches = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    capi.prof_enable(False)
    prof = capi.prof_get()
    tr = d_tr.cpu().numpy()
    corr = d_corr.cpu().numpy()
    ok = np.array_equal(d_pack.cpu().numpy()[corr >= 0][:, :bch.kbch // 8], bb[np.arange(nf) % base][corr >= 0][:, :bch.kbch // 8])
    upd = float(np.where(tr >= 0, tr, args.trials).mean())  # update passes per frame (a batch runs until all of it has converged)
    ms_ldpc = prof.get("k_ldpc_trial", (0.0, 0))[0] / args.steps
    algo = nf * upd * (2 * ldpc.info.msg_bytes_per_frame + 2 * n) + nf * (launches + 1 - upd) * n  # update passes + parity-check-only passes
    out = {"metric": ("DVB-S2 PLFRAME -> BBFRAME frames/s (soft demapper stage + LDPC + repack + BCH), frames resident in HBM" if front else
                      "DVB-S2 FEC frames/s (LDPC + repack + BCH), soft bits resident in HBM"), "value": round(nf / dt, 1), "unit": "frames/s", "coded_Mbit_per_s": round(nf * n / dt / 1e6, 1),
           "config": {"workload": f"{'normal' if args.framesize == 0 else 'short'} FECFRAME rate {args.rate}, {nf} frames per step, "
                                  + (f"QPSK PLFRAMEs at Es/N0 {args.esn0} dB, PLS decoded = {int(d_pls[0])} on every frame: {bool((d_pls == d_pls[0]).all())}, " if front else f"noise sigma {args.sigma} on +-20, ")
                                  + f"max {args.trials} trials, "
                                  f"batch {args.batch} (the reference's SIMD width)"},
           "ms_per_step": round(dt * 1e3, 3), "trial_launches": int(launches) + 1, "update_passes_per_frame": round(upd, 2),
           "frames_converged": float((tr >= 0).mean()), "frames_bch_ok": float((corr >= 0).mean()), "bbframes_match_transmitted": bool(ok),
           "graph": {"layers": ldpc.info.layers, "links": ldpc.info.links_total, "layers_with_shared_bits": ldpc.info.layers_with_shared_bits, "max_phases": ldpc.info.max_phases,
                     "msg_bytes_per_frame": int(ldpc.info.msg_bytes_per_frame)},
           "kernels_ms": {k2: round(v[0] / args.steps, 3) for k2, v in prof.items()},
           "roofline": {"bound": "hbm", "kernel": "k_ldpc_trial", "achieved": round(algo / (ms_ldpc * 1e-3) / 1e9, 1) if ms_ldpc else None, "peak": 8000.0, "unit": "GB/s",
                        "frac": round(algo / (ms_ldpc * 1e-3) / 1e9 / 8000.0, 4) if ms_ldpc else None, "traffic": (_pmc("k_ldpc_trial")[0] * (prof["k_ldpc_trial"][1] / args.steps) if _pmc("k_ldpc_trial")[0] and "k_ldpc_trial" in prof else None),
                        "traffic_per_launch": _pmc("k_ldpc_trial")[0], "launches_per_step": (prof["k_ldpc_trial"][1] / args.steps if "k_ldpc_trial" in prof else None),
                        "traffic_source": _pmc("k_ldpc_trial")[1]}}
    if front and args.sync_frames > 0:
        # the two synchronisation stages in front of the demapper, timed on their own (the frame PLL is one serial lane: it would hide everything else
        # in the step above): PL synchroniser over the frames laid back to back, frame PLL over what it emits
        ns_ = min(nf, args.sync_frames)
        rawlen = 90 + front["nsym"]
        # on-air headers for these stages (SOF + PLS as pi/2-BPSK at the data symbols' amplitude; the frames of the step above carry the header the
        # way the PLL leaves it, which is what the demapper stage reads)
        hdr_air = (np.concatenate([dvbs2_util.sof_symbols(), dvbs2_util.pls_symbols((front["modcod"] << 2) | (args.framesize << 1))]) * a * np.sqrt(2.0)).astype(np.complex64)
        d_true = d_fr.view(nf, front["stride"], 2)[:ns_, :rawlen, :].clone()
        d_true[:, :90, :] = torch.from_numpy(hdr_air.view(np.float32).reshape(90, 2)).cuda()[None, :, :] + sig * torch.randn((ns_, 90, 2), device="cuda", generator=g)
        d_stream = d_true.contiguous().view(-1)
        d_sf = torch.zeros(ns_ * front["stride"] * 2, dtype=torch.float32, device="cuda")
        d_pf = torch.zeros_like(d_sf)
        consumed = C.c_size_t(0)
        bp = np.zeros(ns_, dtype=np.int32)
        slots = front["nsym"] // 90
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nfs = capi.lib().sdhip_s2_pl_sync_dev(0, slots, 0, 0.6, C.c_void_p(d_stream.data_ptr()), ns_ * rawlen, C.c_void_p(d_sf.data_ptr()), front["stride"], ns_, C.byref(consumed),
                                              bp.ctypes.data_as(C.c_void_p))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        lutp = pyref.s2_lut_phase_ref(front["modcod"], args.framesize)
        st2 = np.zeros(2, dtype=np.float32)
        walked = capi.lib().sdhip_s2_pll_dev(0, front["modcod"], args.framesize, 0, 0.002, C.c_void_p(d_sf.data_ptr()), C.c_void_p(d_pf.data_ptr()), front["stride"], int(nfs),
                                             lutp.ctypes.data_as(C.c_void_p), 256, st2.ctypes.data_as(C.c_void_p))
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        # the schedule the MODULE runs (this row used to show the serial lane only): sdhip_s2_pll_frames_dev, mode 2 = a new stream (its first 65 536
        # symbols on the serial lane), then mode 0 on the locked loop's state -- the steady state of a stream
        st3 = np.zeros(2, dtype=np.float32)
        st4 = np.zeros(4, dtype=np.uint32)
        fr_args = (0, front["modcod"], args.framesize, 0, 0.002, C.c_void_p(d_sf.data_ptr()), C.c_void_p(d_pf.data_ptr()), front["stride"], int(nfs), lutp.ctypes.data_as(C.c_void_p), 256)
        capi.lib().sdhip_s2_pll_frames_dev(*fr_args, st3.ctypes.data_as(C.c_void_p), 2, st4.ctypes.data_as(C.c_void_p))
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        walked_p = capi.lib().sdhip_s2_pll_frames_dev(*fr_args, st3.ctypes.data_as(C.c_void_p), 0, st4.ctypes.data_as(C.c_void_p))
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        out["sync_stages"] = {"frames": int(nfs), "symbols": int(nfs) * rawlen, "pl_sync_ms": round((t2 - t1) * 1e3, 3), "pl_sync_Msym_per_s": round(int(nfs) * rawlen / (t2 - t1) / 1e6, 1),
                              "frames_found_at_offset_0": int(np.count_nonzero(bp[:nfs] == 0)),
                              "pll_frame_parallel_ms": round((t5 - t4) * 1e3, 3), "pll_frame_parallel_Msym_per_s": round(int(nfs) * max(walked_p, 0) / (t5 - t4) / 1e6, 2),
                              "pll_frame_parallel_lanes": {"lanes": int(st4[0]), "re_run": int(st4[1]), "forced": int(st4[2]), "serial_frames": int(st4[3])},
                              "pll_serial_lane_ms": round((t3 - t2) * 1e3, 3), "pll_serial_lane_Msym_per_s": round(int(nfs) * walked / (t3 - t2) / 1e6, 2),
                              "pll_state": [float(st2[0]), float(st2[1])],
                              "note": "host wall clock around each call (uploads of the 90 known header symbols and the 256 KB phase-error table included); the module runs the frame-parallel "
                                      "schedule (a locked stream's call, mode 0); the serial lane is the exact mode and a new stream's first 65 536 symbols"}
    if args.cpu_frames > 0:
        m = args.cpu_frames // ref.batch * ref.batch
        if front:  # the decoder's input is what the demapper stage produced
            capi.lib().sdhip_s2_bb_to_soft_dev(0, front["modcod"], args.framesize, 0, C.c_void_p(d_fr.data_ptr()), front["stride"], m, lut.ctypes.data_as(C.c_void_p), 256,
                                               C.c_void_p(soft.data_ptr()), None)
        sh = soft[:m].cpu().numpy()
        t1 = time.perf_counter()
        want, wt = ref.ldpc_decode(args.framesize, rc, sh, args.trials)
        t2 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(m / (t2 - t1), 1), "unit": "frames/s (LDPC only)", "cores": 1, "kind": "reference",
                               "sample": f"first {m} frames, BBFrameLDPC::decode, SIMD width {ref.batch}"}
        if ref.batch == args.batch:
            out["parity_sample"] = {"frames": m, "soft_bits_identical": bool(np.array_equal(work[:m].cpu().numpy(), want)), "trials_identical": bool(np.array_equal(tr[:m // ref.batch], wt))}
    return out


def _pmc(kernel):
    """HBM bytes per launch of `kernel` from the PMC profile of THIS bench, when one was committed for these kernel sources (bench.pmc_traffic, tag dvbs2fec)"""
    import bench as _b
    return _b.pmc_traffic("dvbs2fec", kernel)


def main():
    print(json.dumps(run(parse())), flush=True)


if __name__ == "__main__":
    main()
