#!/usr/bin/env python3
"""ndsp costas_fast_cc (satdump::ndsp::CostasFastBlock, dsp/pll/costas_fast.cpp) on one MI355X, samples resident in HBM: the loop lane-per-chunk
with the STRICT hand-off (a chunk stands only if its start state is bit-identical to its predecessor's end state modulo an exact quarter turn: DemodEngine::costas_fast_stage),
so the output is the reference block's float for float -- checked here over the WHOLE first call against the block compiled in place (oracle/_ref), whose own rate on one
host thread is the CPU figure.   usage: tools/bench_costas_fast.py [--samples 268435456] [--steps 5]"""
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
    from satdump_amd import capi

    # QPSK symbols at one sample per symbol (what the hier block's clock recovery hands its carrier loop): carrier offset, noise; generated on the device, copied back for the reference
    n = args.samples
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    bits = torch.randint(0, 4, (n,), device="cuda", generator=g)
    ang = (bits.to(torch.float32) * 0.5 + 0.25) * np.float32(np.pi)
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    ph = (ang.to(torch.float64) + 0.02 * t + 0.3).to(torch.float64)
    amp = 0.6
    x = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    x[:, 0] = (amp * torch.cos(ph)).to(torch.float32)
    x[:, 1] = (amp * torch.sin(ph)).to(torch.float32)
    del ph, t, ang, bits
    x += 0.12 * torch.randn((n, 2), device="cuda", generator=g, dtype=torch.float32)  # ~11 dB
    d_x = x.reshape(-1).contiguous()
    d_y = torch.zeros(2 * n + 64, dtype=torch.float32, device="cuda")

    L = capi.lib()
    c = capi.NdspPskCfg()
    L.sdhip_ndsp_psk_cfg_default(C.byref(c))
    c.constellation = capi.QPSK
    c.pll_loop_bw = args.loop_bw
    KIND_COSTAS_FAST = 7
    h = L.sdhip_ndsp_block_create(KIND_COSTAS_FAST, C.byref(c))
    assert h, capi.last_error()

    def step():
        r = L.sdhip_ndsp_psk_demod_work_dev(h, C.c_void_p(d_x.data_ptr()), n, C.c_void_p(d_y.data_ptr()), n + 32)
        assert r == n, capi.last_error()

    def stats():
        st = capi.DemodStats()
        L.sdhip_ndsp_psk_demod_get_stats(h, C.byref(st))
        return dict(chunks=st.chunks, re_run=st.chunks_fixed, sequential_fallbacks=st.chunks_forced)

    t0 = time.time()
    step()
    torch.cuda.synchronize()
    first_ms = (time.time() - t0) * 1e3
    first = stats()
    got = d_y[: 2 * n].cpu().numpy().view(np.uint32)
    xh = d_x.cpu().numpy().view(np.complex64)
    nref = pyref.NdspRef()
    t0 = time.time()
    want = nref.run("costas_fast_cc", {"order": 4, "loop_bw": args.loop_bw}, xh, buf=1 << 16)
    t_cpu = time.time() - t0
    identical = bool(np.array_equal(got, want.view(np.uint32)))
    ndiff = int(np.count_nonzero(got != want.view(np.uint32))) if not identical else 0
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
    return {"row": "ndsp costas_fast_cc (dsp/pll/costas_fast.cpp), QPSK symbols at 1 sample per symbol, loop_bw %g" % args.loop_bw, "samples_per_call": n,
            "value": round(n / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_call": round(dt * 1e3, 3), "first_call_ms": round(first_ms, 2),
            "first_call": first, "steady_call": steady,
            "bit_identical_to_the_reference_block": identical, "samples_compared": n, "words_differing": ndiff,
            "cpu_reference": {"value": round(n / t_cpu / 1e6, 1), "unit": "Msamples/s", "cores": 1, "kind": "reference", "sample": "the whole first call, the block on its own thread between two FIFOs"},
            "algo_bytes_per_call": 16 * n, "whole_GBps": round(16 * n / dt / 1e9, 1), "kernels_ms_per_call": kern}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1 << 28)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loop-bw", dest="loop_bw", type=float, default=0.004)
    print(json.dumps(run(ap.parse_args())))
