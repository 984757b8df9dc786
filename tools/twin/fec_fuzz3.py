"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
MetOp AHRPT decoder and the punctured conv_rate path with random noise levels, bursts and push boundaries (minutes per case on the twin).
Usage: python tools/twin/fec_fuzz3.py <seed> <iterations>   (from the repository root)"""
import sys, os, importlib.util, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT)
import numpy as np
from tests.emu import build as emu_build
from oracle import pyref
from satdump_amd import synth
from tests import util
lib=emu_build.build()
os.environ["SDHIP_TESTING_TWIN"]="1"; os.environ["SDHIP_LIB"]=lib
spec=importlib.util.spec_from_file_location("capi_emu",os.path.join(ROOT,"satdump_amd","capi.py")); twin=importlib.util.module_from_spec(spec); spec.loader.exec_module(twin); twin.lib()
del os.environ["SDHIP_LIB"]
orc=pyref.best()
rng=np.random.default_rng(int(sys.argv[1])); bad=0
for it in range(int(sys.argv[2])):
    t=time.time()
    if rng.random()<0.5:
        sp, cadus, plain, syms = util.metop_case(nframes=int(rng.integers(5,9)), seed=int(rng.integers(1<<20)))
        soft=synth.soft_from_symbols(syms, sp, sigma=float(rng.choice([20,40,60])), seed=int(rng.integers(1<<20)))
        if rng.random()<0.5:
            p=int(rng.integers(0,len(soft)-4000)); soft[p:p+int(rng.integers(100,3000))]=0
        soft=soft[:len(soft)//16384*16384]
        want=orc.metop_decode(soft)["cadu"]
        dec=twin.FecDecoder(twin.fec_cfg(decoder=1, viterbi_ber_thresold=0.17, viterbi_outsync_after=5)); name="metop"
    else:
        rate=int(rng.integers(1,5)); const=str(rng.choice(["qpsk","bpsk"])); nrzm=int(rng.random()<0.5)
        soft,plain=util.punctured_case(rate,nframes=int(rng.integers(4,8)),sigma=float(rng.choice([8,14])),seed=int(rng.integers(1<<20)),nrzm=bool(nrzm),gap=bool(rng.random()<0.4),prefix=int(rng.integers(0,5000)))
        want=orc.concat_decode_punc(pyref.fec_cfg(constellation={"qpsk":pyref.QPSK,"bpsk":pyref.BPSK}[const],nrzm=nrzm,rs_usecheck=1),rate,soft)["cadu"]
        dec=twin.FecDecoder(twin.fec_cfg(constellation=const,nrzm=nrzm,rs_i=4,rs_type=1,rs_usecheck=1,conv_rate=rate)); name=f"punct rate {rate} {const} nrzm {nrzm}"
    cuts=sorted(set([0,len(soft)]+rng.integers(0,len(soft),int(rng.integers(0,4))).tolist()))
    got=[]
    for a,b in zip(cuts[:-1],cuts[1:]):
        dec.push(soft[a:b]); got.append(dec.pull())
    got=np.concatenate(got)
    ok=got.shape==want.shape and np.array_equal(got,want)
    print(it,name,"pushes",len(cuts)-1,"cadus",len(got),"/",len(want),"OK" if ok else "MISMATCH",round(time.time()-t,1),"s",flush=True)
    bad+=not ok
print("bad",bad)
