"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
chunk-parallel mode: per-stage lane targets, explicit warm-ups, tiny calls, cs16 through push/pull and process_dev.
Usage: python tools/twin/emu_fuzz3.py <seed> <iterations>   (from the repository root)"""
import sys, os, importlib.util, time
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_demod_emu_cpu as T
from satdump_amd import synth
from oracle import pyref
lib=T.emu_build.build()
os.environ["SDHIP_TESTING_TWIN"]="1"; os.environ["SDHIP_LIB"]=lib
spec=importlib.util.spec_from_file_location("capi_emu",os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),"satdump_amd","capi.py")); twin=importlib.util.module_from_spec(spec); spec.loader.exec_module(twin); twin.lib()
del os.environ["SDHIP_LIB"]
orc=pyref.best()
rng=np.random.default_rng(int(sys.argv[1]))
cases={}
for c in ("goes","metop","npp"):
    plain, x, ocfg, kw, ofec = T._case(c, 40)
    s16=synth.to_cs16(x); xf=(s16.astype(np.float32)*np.float32(1.0/32767.0)).view(np.complex64)
    want=orc.psk_demod(ocfg,x); want16=orc.psk_demod(ocfg,xf)
    cases[c]=(x,s16,ocfg,kw,ofec,want,T._cadus(orc,c,ofec,want["soft"]),want16,T._cadus(orc,c,ofec,want16["soft"]))
bad=0
for it in range(int(sys.argv[2])):
    c=str(rng.choice(list(cases))); x,s16,ocfg,kw,ofec,want,wantc,want16,wantc16=cases[c]
    n=len(x)
    mode=str(rng.choice(["lanes","warmup","tiny","push16","dev16"]))
    env={}; extra={}
    cuts=sorted(set([0,n]+rng.integers(0,n,int(rng.integers(0,5))).tolist()))
    if mode=="lanes":
        env={"SDHIP_LANES_AGC":str(int(rng.integers(20,200))),"SDHIP_LANES_COSTAS":str(int(rng.integers(20,300))),"SDHIP_LANES_MM":str(int(rng.integers(20,200)))}
    elif mode=="warmup":
        extra=dict(chunk_len=int(rng.choice([4096,8192])), warmup=int(rng.choice([2048,4096,8192])))
    elif mode=="tiny":
        extra=dict(chunk_len=4096)
        pos=sorted(rng.integers(0,n,3).tolist()); cuts=[0]
        for p in pos: cuts += [p, p+int(rng.integers(1,300)), p+int(rng.integers(300,3000))]
        cuts=sorted(set([c_ for c_ in cuts if c_<n]+[n]))
    else:
        extra=dict(chunk_len=int(rng.choice([4096,8192])))
    if rng.random()<0.3: env["SDHIP_CKPT"]="1"
    for k,v in env.items(): os.environ[k]=v
    try:
        if mode=="push16":
            dem=twin.PskDemod(twin.demod_cfg(**kw,**extra))
            for a,b in zip(cuts[:-1],cuts[1:]): dem.push(s16[2*a:2*b], twin.FMT_CS16)
            dem.flush(); soft=dem.pull(); dem.close(); w,wc=want16,wantc16; nsym=len(soft)
            ok_n = len(soft)==len(w["soft"])
        elif mode=="dev16":
            cfg=twin.demod_cfg(**kw,**extra); dem=twin.PskDemod(cfg); outs=[]
            for a,b in zip(cuts[:-1],cuts[1:]):
                m=b-a; o=np.zeros(2*m+64,dtype=np.int8); seg=np.ascontiguousarray(s16[2*a:2*b])
                ns=dem.process_dev(seg.ctypes.data, m, twin.FMT_CS16, o.ctypes.data, 2*m+64); outs.append(o[:ns].copy())
            soft=np.concatenate(outs); dem.close(); w,wc=want16,wantc16; ok_n=len(soft)==len(w["soft"])
        else:
            soft,syms,st=T._run(twin,kw,x,chunks=cuts,**extra); w,wc=want,wantc; ok_n=len(syms)==len(w["syms"])
    except Exception as e:
        print(it,c,mode,"EXCEPTION",e); bad+=1
        for k in env: del os.environ[k]
        continue
    for k in env: del os.environ[k]
    got=T._cadus(orc,c,ofec,soft)
    ok=ok_n and got.shape==wc.shape and np.array_equal(got,wc)
    print(it,c,mode,extra,env,"calls",len(cuts)-1,"cadus",len(got),"/",len(wc),"OK" if ok else "MISMATCH", "" if ok_n else "NSYM")
    bad+=not ok
print("bad",bad)
