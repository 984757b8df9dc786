"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
FEC engine: frame formats (interleaving depth, RS 223/239, dual basis, derand after RS, fill bytes, no RS) with injected byte errors.
Usage: python tools/twin/fec_fuzz2.py <seed> <iterations>   (from the repository root)"""
import sys, os, importlib.util, time
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.emu import build as emu_build
from oracle import pyref
from satdump_amd import synth
lib=emu_build.build()
os.environ["SDHIP_TESTING_TWIN"]="1"; os.environ["SDHIP_LIB"]=lib
spec=importlib.util.spec_from_file_location("capi_emu",os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),"satdump_amd","capi.py")); twin=importlib.util.module_from_spec(spec); spec.loader.exec_module(twin); twin.lib()
del os.environ["SDHIP_LIB"]
orc=pyref.best()
rng=np.random.default_rng(int(sys.argv[1])); bad=0
for it in range(int(sys.argv[2])):
    I=int(rng.choice([4,5,1,2])); t239=int(rng.random()<0.3); dual=int(rng.random()<0.7); after=int(rng.random()<0.3)
    fill=int(rng.choice([-1,-1,0,3])); usecheck=int(rng.random()<0.6); nors=int(rng.random()<0.15)
    nroots=16 if t239 else 32
    nfr=int(rng.integers(4,9))
    cadus=synth.make_cadus(nfr, seed=int(rng.integers(1<<30)), rs_i=I, dualbasis=bool(dual), derand=True, nroots=nroots)
    # a few byte errors in some frames (correctable and not)
    for f in range(nfr):
        ne=int(rng.choice([0,0,3,12,40]))
        for p in rng.choice(cadus.shape[1]-4, ne, replace=False): cadus[f,4+p]^=int(rng.integers(1,256))
    sp=synth.SynthSpec(constellation="bpsk", samplerate=3e6, symbolrate=1e6, nrzm=True, seed=int(rng.integers(1<<30)))
    syms=synth.frames_to_symbols(cadus, sp)
    soft=synth.soft_from_symbols(syms, sp, sigma=15.0, seed=int(rng.integers(1<<30)))
    cs=cadus.shape[1]*8
    bufsz=max(cs,8192)
    soft=np.concatenate([soft, rng.integers(-127,128,bufsz).astype(np.int8)]); soft=soft[:len(soft)//bufsz*bufsz]
    kw=dict(nrzm=1, rs_usecheck=usecheck, cadu_size=cs, rs_i=0 if nors else I, rs_type=(2 if t239 else 1), rs_dualbasis=dual, derand_after_rs=after, rs_fill_bytes=fill)
    okw=dict(kw); okw["rs_type"]=pyref.RS239 if t239 else pyref.RS223
    try:
        want=orc.concat_decode(pyref.fec_cfg(constellation=pyref.BPSK, **okw), soft)
    except Exception as e:
        print("oracle error",e); continue
    t=time.time()
    try:
        dec=twin.FecDecoder(twin.fec_cfg(constellation="bpsk", **kw)); dec.push(soft); got=dec.pull()
    except Exception as e:
        print(it,"TWIN ERROR",kw,e); bad+=1; continue
    ok=got.shape==want["cadu"].shape and np.array_equal(got,want["cadu"])
    print(it,kw,"cadus",len(got),"/",len(want["cadu"]),"OK" if ok else "MISMATCH",round(time.time()-t,1),"s")
    bad+=not ok
print("bad",bad)
