"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
FEC engine: random constellation / NRZ-M / derandomiser / RS check, garbage prefix, bursts, polarity inversion, ragged pushes.
Usage: python tools/twin/fec_fuzz.py <seed> <iterations>   (from the repository root)"""
import sys, os, importlib.util, time
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.emu import build as emu_build
from oracle import pyref
from satdump_amd import synth
from tests import util
lib=emu_build.build()
os.environ["SDHIP_TESTING_TWIN"]="1"; os.environ["SDHIP_LIB"]=lib
spec=importlib.util.spec_from_file_location("capi_emu",os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),"satdump_amd","capi.py")); twin=importlib.util.module_from_spec(spec); spec.loader.exec_module(twin); twin.lib()
del os.environ["SDHIP_LIB"]
orc=pyref.best()
rng=np.random.default_rng(int(sys.argv[1])); bad=0
for it in range(int(sys.argv[2])):
    const=str(rng.choice(["bpsk","qpsk","oqpsk","bpsk_90"])) if False else str(rng.choice(["bpsk","qpsk","oqpsk"]))
    nrzm=int(rng.random()<0.6); usecheck=int(rng.random()<0.7); derand=int(rng.random()<0.85); iq_inv=int(rng.random()<0.2)
    nfr=int(rng.integers(4,10)); sigma=float(rng.choice([10,20,30]))
    sp=synth.SynthSpec(constellation="qpsk" if const!="bpsk" else "bpsk", samplerate=3e6, symbolrate=1e6, nrzm=bool(nrzm), seed=int(rng.integers(1<<30)))
    cadus=synth.make_cadus(nfr, seed=int(rng.integers(1<<30)), derand=bool(derand))
    syms=synth.frames_to_symbols(cadus, sp)
    soft=synth.soft_from_symbols(syms, sp, sigma=sigma, seed=int(rng.integers(1<<30)))
    # disturbances: garbage prefix, a burst, an inversion
    pre=rng.integers(-127,128,int(rng.integers(0,3000))).astype(np.int8)
    soft=np.concatenate([pre,soft])
    if rng.random()<0.5:
        p=int(rng.integers(0,len(soft)-3000)); soft[p:p+int(rng.integers(50,2500))]=rng.integers(-127,128,1).astype(np.int8)[0]
    if rng.random()<0.3:
        h=len(soft)//2; soft[h:]=(-soft[h:].astype(np.int16)).clip(-127,127).astype(np.int8)
    soft=np.concatenate([soft, rng.integers(-127,128,8192).astype(np.int8)])
    soft=soft[:len(soft)//8192*8192]
    oc=dict(constellation={"bpsk":pyref.BPSK,"qpsk":pyref.QPSK,"oqpsk":pyref.OQPSK}[const], nrzm=nrzm, rs_usecheck=usecheck, derandomize=derand, iq_invert=iq_inv)
    want=orc.concat_decode(pyref.fec_cfg(**oc), soft)
    cfg=twin.fec_cfg(constellation=const, nrzm=nrzm, rs_i=4, rs_type=1, rs_usecheck=usecheck, derandomize=derand, iq_invert=iq_inv)
    t=time.time()
    dec=twin.FecDecoder(cfg)
    # ragged pushes
    cuts=sorted(set([0,len(soft)]+rng.integers(0,len(soft),int(rng.integers(0,4))).tolist()))
    got=[]
    for a,b in zip(cuts[:-1],cuts[1:]):
        dec.push(soft[a:b]); got.append(dec.pull())
    got=np.concatenate(got) if got else np.zeros((0,1024),np.uint8)
    ber,st=dec.block_taps()
    ok=got.shape==want["cadu"].shape and np.array_equal(got,want["cadu"])
    print(it,const,"nrzm",nrzm,"chk",usecheck,"derand",derand,"inv",iq_inv,"frames",nfr,"sigma",sigma,"pushes",len(cuts)-1,"cadus",len(got),"/",len(want["cadu"]),"OK" if ok else "MISMATCH", round(time.time()-t,1),"s")
    bad+=not ok
print("bad",bad)
