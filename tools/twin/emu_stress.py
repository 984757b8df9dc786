"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
disturbed streams (noise gap, level step, frequency step, leading noise) on the three workloads.
Usage: python tools/twin/emu_stress.py <seed> <iterations>   (from the repository root)"""
import sys, os, time, importlib.util, ctypes as C
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_demod_emu_cpu as T
from oracle import pyref
lib=T.emu_build.build()
os.environ["SDHIP_TESTING_TWIN"]="1"; os.environ["SDHIP_LIB"]=lib
spec=importlib.util.spec_from_file_location("capi_emu",os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),"satdump_amd","capi.py")); twin=importlib.util.module_from_spec(spec); spec.loader.exec_module(twin); twin.lib()
del os.environ["SDHIP_LIB"]
orc=pyref.best()
def report(tag, case, x, ocfg, kw, ofec, **extra):
    want = orc.psk_demod(ocfg, x); wantc=T._cadus(orc,case,ofec,want["soft"])
    soft, syms, st = T._run(twin, kw, x, **extra)
    got=T._cadus(orc,case,ofec,soft)
    ws={bytes(c) for c in wantc}; gs={bytes(c) for c in got}
    print(f"{tag:34s} chunks {st.chunks:5d} fixed {st.chunks_fixed:4d} forced {st.chunks_forced:4d} nsym {len(syms)}/{len(want['syms'])} ref cadus {len(wantc):3d} ours {len(got):3d} common {len(ws&gs):3d} only_ref {len(ws-gs)} only_ours {len(gs-ws)}")
rng=np.random.default_rng(1)
for case in ("npp","goes","metop"):
    plain, x, ocfg, kw, ofec = T._case(case, 60)
    n=len(x)
    sig=np.std(x)
    noise=lambda m: ((rng.standard_normal(m)+1j*rng.standard_normal(m))*sig/np.sqrt(2)).astype(np.complex64)
    # 1. noise gap in the middle
    g=n//2
    xg=np.concatenate([x[:g], noise(150000), x[g:]])
    report(case+" noise gap", case, xg, ocfg, kw, ofec, chunk_len=4096)
    # 2. amplitude step x4 and back
    xa=x.copy(); xa[n//3:2*n//3]*=4
    report(case+" amplitude step", case, xa, ocfg, kw, ofec, chunk_len=4096)
    # 3. frequency step (+0.002 rad/sample) at the middle
    xf=x.copy(); k=np.arange(n-n//2); xf[n//2:]*=np.exp(1j*0.002*k).astype(np.complex64)
    report(case+" frequency step", case, xf, ocfg, kw, ofec, chunk_len=4096)
    # 4. stream starts with noise
    xs=np.concatenate([noise(100000), x])
    report(case+" noise first", case, xs, ocfg, kw, ofec, chunk_len=4096)
    # 5. many small calls
    b=[0]+sorted(rng.integers(1,n,12).tolist())+[n]
    report(case+" 13 ragged calls", case, x, ocfg, kw, ofec, chunk_len=4096, chunks=b) if False else None
