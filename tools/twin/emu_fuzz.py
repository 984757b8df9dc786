"""Differential fuzz of the HOST TWIN (tests/emu) against the oracle -- development tool, not part of the test suite.
exact mode, random sample/symbol rates (resampler or not), taps, loop gains, dc_block, iq_swap, call splits: must be bit-identical to the oracle.
Usage: python tools/twin/emu_fuzz.py <seed> <iterations>   (from the repository root)"""
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
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
CN={"bpsk":pyref.BPSK,"qpsk":pyref.QPSK,"oqpsk":pyref.OQPSK,"8psk":pyref.PSK8}
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 30):
    const=rng.choice(["bpsk","qpsk","qpsk","oqpsk","8psk"])
    symrate=float(rng.choice([927000, 2333333, 665400, 15e6, 3.5e6]))
    sps=float(rng.uniform(1.2, 7.5))
    fs=round(symrate*sps/1000)*1000.0
    alpha=float(rng.choice([0.35,0.5,0.6])); ntaps=int(rng.choice([31,31,51,21]))
    pll=float(rng.choice([0.002,0.003,0.006,0.02])); agc=float(rng.choice([1e-2,1e-3,1e-4]))
    dc=int(rng.random()<0.2); swap=int(rng.random()<0.2)
    n=int(rng.integers(20000,120000))
    # signal: QPSK-ish random symbols shaped, or plain noise (arithmetic parity does not need a real signal)
    sp=synth.SynthSpec(constellation="qpsk" if const!="bpsk" else "bpsk", samplerate=fs, symbolrate=symrate, seed=int(rng.integers(1<<30)))
    nsym=int(n/ (fs/symrate))+64
    syms=rng.integers(0,2,(nsym,2 if const!="bpsk" else 1)).astype(np.uint8)
    try:
        x,_=synth.modulate(syms.reshape(-1) if const=="bpsk" else syms, sp)
    except Exception as e:
        x=((rng.standard_normal(n)+1j*rng.standard_normal(n))*0.3).astype(np.complex64)
    x=np.ascontiguousarray(x[:n].astype(np.complex64))
    if len(x)<n: n=len(x)
    kwo=dict(samplerate=fs,symbolrate=symrate,constellation=CN[const],rrc_alpha=alpha,rrc_taps=ntaps,pll_bw=pll,agc_rate=agc,dc_block=dc,iq_swap=swap)
    kwt=dict(samplerate=fs,symbolrate=symrate,constellation=const,rrc_alpha=alpha,rrc_taps=ntaps,pll_bw=pll,agc_rate=agc,dc_block=dc,iq_swap=swap)
    try:
        want=orc.psk_demod(pyref.demod_cfg(**kwo), x)
    except Exception as e:
        print("oracle refused", kwo, e); continue
    cuts=sorted(set([0,n]+rng.integers(0,n,int(rng.integers(0,5))).tolist()))
    try:
        soft,syms_,st=T._run(twin,kwt,x,chunks=cuts,exact=1)
    except Exception as e:
        print("TWIN ERROR", kwt, n, cuts, e); bad+=1; continue
    ok=len(soft)==len(want["soft"]) and np.array_equal(soft,want["soft"]) and np.array_equal(syms_.view(np.uint32),want["syms"].view(np.uint32)) and st.buffer_size==want["buffer_size"]
    # chunked mode: symbol count and sanity
    soft2,syms2,st2=T._run(twin,kwt,x,chunk_len=int(rng.choice([2048,4096,8192])))
    ok2=abs(len(syms2)-len(want["syms"]))<=max(2, 0.002*len(syms2)) if st2.chunks_forced==0 else True
    print(it,const,"fs %.0f sps %.3f"%(fs,fs/symrate),"taps",ntaps,"pll",pll,"agc",agc,"dc",dc,"swap",swap,"n",n,"calls",len(cuts)-1,"final_sps %.4f"%want["final_sps"],"EXACT",ok,"| chunked nsym diff",len(syms2)-len(want["syms"]),"forced",st2.chunks_forced,"fixed",st2.chunks_fixed,"OK" if ok2 else "MISMATCH")
    if not ok or not ok2: bad+=1
print("bad",bad)
