#!/usr/bin/env python3
"""k_rs_screen's launch time on N clean CADUs resident in HBM (RS(255,223) I = 4, dual basis): the library named by SDHIP_LIB (default: the product's).
EXPERIMENT TOOL: A/B of the packed screen of branch r5/rs-screen-perm (tools/build_variant.sh-style library) against the per-codeword kernel."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import torch
    from satdump_amd import capi, synth
    base = synth.make_cadus(1024, seed=11, derand=False)
    d = torch.from_numpy(np.ascontiguousarray(base)).cuda().repeat((n + 1023) // 1024, 1)[:n].contiguous()
    err = torch.zeros(n * 4, dtype=torch.int32, device="cuda")
    L = capi.lib()
    for _ in range(3):
        rc = L.sdhip_op_rs_decode(0, C.c_void_p(d.data_ptr() + 4), n, 1024, 1, 4, capi.RS223, 0, C.c_void_p(err.data_ptr()))
        assert rc == 0, capi.last_error()
    torch.cuda.synchronize()
    capi.prof_reset()
    capi.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        L.sdhip_op_rs_decode(0, C.c_void_p(d.data_ptr() + 4), n, 1024, 1, 4, capi.RS223, 0, C.c_void_p(err.data_ptr()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    capi.prof_enable(False)
    prof = capi.prof_get()
    print({"lib": os.path.basename(capi.LIB_PATH), "frames": n, "errors_nonzero": int((err != 0).sum().item()), "ms_per_call": round(dt * 1e3, 4),
           "kernels_ms_per_call": {k: round(v[0] / reps, 4) for k, v in prof.items()}}, flush=True)


if __name__ == "__main__":
    main()
