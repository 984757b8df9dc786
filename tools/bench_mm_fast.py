#!/usr/bin/env python3
"""ndsp fast_clock_recovery_mm_cc (satdump::ndsp::MMClockRecoveryFastBlock<complex_t>, dsp/clock_recovery/clock_recovery_mm_fast.cpp) on one MI355X, samples
resident in HBM: a lane per (chunk, cadence of the block's every-fifth-symbol rate update) with the STRICT hand-off (DemodEngine::mmfast_stage, k_mmfast) -- the output is the
reference block's float for float, checked here over the WHOLE first call against the block compiled in place (oracle/_ref), whose own rate on one host thread is the CPU figure.
Input: QPSK at three samples per symbol behind the HIP rrc_fir_cc and agc_cc blocks (made on the device, copied back for the reference).
usage: tools/bench_mm_fast.py [--samples 134217728] [--steps 3]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args) -> dict:
    import torch
    torch.zeros(1, device="cuda")
    from oracle import pyref
    from satdump_amd import capi, ndsp, synth

    nsym_blk = 1 << 20
    rng = np.random.default_rng(13)
    a = ((rng.integers(0, 2, nsym_blk) * 2.0 - 1.0) + 1j * (rng.integers(0, 2, nsym_blk) * 2.0 - 1.0)) / np.sqrt(2.0)
    spec = synth.SynthSpec(constellation="qpsk", samplerate=6e6, symbolrate=2e6, rrc_alpha=0.35, amplitude=0.4, cfo_hz=0.0, esn0_db=10.0, seed=13)
    blk, _ = synth.modulate(a, spec, periodic=True)
    reps = max(1, args.samples // len(blk))
    n0 = reps * len(blk)
    d_raw = torch.from_numpy(blk.view(np.float32)).cuda().repeat(reps)
    d_f = torch.zeros(2 * n0 + 64, dtype=torch.float32, device="cuda")
    d_x = torch.zeros(2 * n0 + 64, dtype=torch.float32, device="cuda")
    fir = ndsp.SingleBlock("rrc_fir_cc")
    for k, v in {"samplerate": 6e6, "symbolrate": 2e6, "alpha": 0.35}.items():
        fir.set_cfg(k, v)
    nf = fir.work_dev(d_raw.data_ptr(), n0, d_f.data_ptr(), n0 + 32)
    agc = ndsp.SingleBlock("agc_cc")
    for k, v in {"rate": 1e-3, "reference": 0.6}.items():
        agc.set_cfg(k, v)
    n = agc.work_dev(d_f.data_ptr(), nf, d_x.data_ptr(), n0 + 32)
    fir.stop()
    agc.stop()
    del d_raw, d_f
    d_y = torch.zeros(2 * (n // 2 + 64), dtype=torch.float32, device="cuda")

    L = capi.lib()
    c = capi.NdspPskCfg()
    L.sdhip_ndsp_psk_cfg_default(C.byref(c))
    c.rec_omega = 3.0
    KIND_MM_FAST = 8
    h = L.sdhip_ndsp_block_create(KIND_MM_FAST, C.byref(c))
    assert h, capi.last_error()

    def step():
        r = L.sdhip_ndsp_psk_demod_work_dev(h, C.c_void_p(d_x.data_ptr()), n, C.c_void_p(d_y.data_ptr()), n // 2 + 32)
        assert r > 0, capi.last_error()
        return r

    def stats():
        st = capi.DemodStats()
        L.sdhip_ndsp_psk_demod_get_stats(h, C.byref(st))
        return dict(chunks=st.chunks, re_run=st.chunks_fixed, sequential_fallbacks=st.chunks_forced)

    t0 = time.time()
    ns = step()
    torch.cuda.synchronize()
    first_ms = (time.time() - t0) * 1e3
    first = stats()
    got = d_y[: 2 * ns].cpu().numpy().view(np.uint32)
    xh = d_x[: 2 * n].cpu().numpy().view(np.complex64)
    nref = pyref.NdspRef()
    t0 = time.time()
    want = nref.run("fast_clock_recovery_mm_cc", {"omega": 3.0}, xh, buf=1 << 16)
    t_cpu = time.time() - t0
    identical = bool(len(want) == ns and np.array_equal(got, want.view(np.uint32)))
    ndiff = -1 if len(want) != ns else int(np.count_nonzero(got != want.view(np.uint32)))
    del got, want, xh

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    capi.prof_reset()
    capi.prof_enable(True)
    t0 = time.time()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    capi.prof_enable(False)
    prof = capi.prof_get()
    steady = stats()
    L.sdhip_ndsp_psk_demod_destroy(h)
    kern = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}
    return {"row": "ndsp fast_clock_recovery_mm_cc (dsp/clock_recovery/clock_recovery_mm_fast.cpp), QPSK at 3 samples per symbol behind rrc_fir_cc + agc_cc, the block's default gains",
            "samples_per_call": n, "symbols_per_call": int(ns), "value": round(n / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_call": round(dt * 1e3, 3), "first_call_ms": round(first_ms, 2),
            "first_call": first, "steady_call": steady, "bit_identical_to_the_reference_block": identical, "symbols_compared": int(ns), "words_differing": ndiff,
            "cpu_reference": {"value": round(n / t_cpu / 1e6, 1), "unit": "Msamples/s", "cores": 1, "kind": "reference", "sample": "the whole first call, the block on its own thread between two FIFOs"},
            "algo_bytes_per_call": 8 * n + 8 * int(ns), "kernels_ms_per_call": kern}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1 << 27)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    print(json.dumps(run(ap.parse_args())))
