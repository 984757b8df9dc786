"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
chunk-parallel mode on the three workloads, random chunk lengths / call splits / cut warm-ups / SDHIP_CKPT: CADUs must equal the reference chain's.
Usage: python tools/twin/emu_fuzz2.py <seed> <iterations>   (from the repository root)"""
import sys, os, importlib.util, time
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_demod_emu_cpu as T
from oracle import pyref
lib=T.emu_build.build()
os.environ["SDHIP_TESTING_TWIN"]="1"; os.environ["SDHIP_LIB"]=lib
spec=importlib.util.spec_from_file_location("capi_emu",os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),"satdump_amd","capi.py")); twin=importlib.util.module_from_spec(spec); spec.loader.exec_module(twin); twin.lib()
del os.environ["SDHIP_LIB"]
orc=pyref.best()
rng=np.random.default_rng(int(sys.argv[1]))
cases={}
for c in ("goes","metop","npp"):
    plain, x, ocfg, kw, ofec = T._case(c, 50)
    want=orc.psk_demod(ocfg,x); cases[c]=(x,ocfg,kw,ofec,want,T._cadus(orc,c,ofec,want["soft"]))
bad=0
for it in range(int(sys.argv[2])):
    c=str(rng.choice(list(cases))); x,ocfg,kw,ofec,want,wantc=cases[c]
    n=len(x)
    cl=int(rng.choice([2048,3072,4096,6144,8192,16384]))
    cuts=sorted(set([0,n]+rng.integers(0,n,int(rng.integers(0,6))).tolist()))
    env={}
    if rng.random()<0.5: env["SDHIP_CKPT"]="1"
    if rng.random()<0.3: env["SDHIP_W_MM"]=str(int(rng.choice([512,1024])))
    if rng.random()<0.2: env["SDHIP_W_COSTAS"]=str(int(rng.choice([512,1024,2048])))
    for k,v in env.items(): os.environ[k]=v
    soft,syms,st=T._run(twin,kw,x,chunks=cuts,chunk_len=cl)
    for k in env: del os.environ[k]
    got=T._cadus(orc,c,ofec,soft)
    ok=len(syms)==len(want["syms"]) and got.shape==wantc.shape and np.array_equal(got,wantc)
    print(it,c,"cl",cl,"calls",len(cuts)-1,env,"chunks",st.chunks,"fixed",st.chunks_fixed,"forced",st.chunks_forced,"nsym diff",len(syms)-len(want["syms"]),"cadus",len(got),"/",len(wantc),"OK" if ok else "MISMATCH")
    bad+=not ok
print("bad",bad)
