# study tool (not product, not test): warm-up gear shift of the M&M loop. Build the helper first:
#   gcc -O2 -shared -fPIC -o /tmp/libmm_gear_study.so tools/mm_gear_study.c -lm
import sys; sys.path.insert(0, '/root/repo')
import ctypes as C, numpy as np
from oracle import pyref
from satdump_amd import synth
from tests import util
P = pyref.port()
L = C.CDLL('/tmp/libmm_gear_study.so')
L.mm_track.restype = C.c_long
L.mm_track.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_long, C.c_float, C.c_float, C.c_long, C.c_void_p, C.c_void_p, C.c_int]
bank = np.ascontiguousarray(P.mm_bank(128, 8), dtype=np.float32)
def track(c, start, sps, nfast, G, Gom, ntot, freeze=0, og=(8.7e-3)**2/4, mg=8.7e-3):
    t = np.zeros(ntot); om = np.zeros(ntot, dtype=np.float32)
    m = L.mm_track(c.ctypes.data, len(c), start, bank.ctypes.data, sps, og, mg, 0.005, nfast, G, Gom, ntot, t.ctypes.data, om.ctypes.data, freeze)
    return t[:m], om[:m]
def case(name):
    if name == "goes":
        spec, cadus, plain, syms = util.goes_case(nframes=30); sps = 2.9126; bw = 0.02; order = 2
        x, _ = synth.modulate(syms, spec); x = P.block(4, [2700000, 3000000], x); fs, sr = 2.7e6, 927000
    elif name == "metop":
        spec, cadus, plain, syms = util.metop_case(nframes=50); sps = 2.5714290142; bw = 0.003; order = 4
        x, _ = synth.modulate(syms, spec); fs, sr = 6e6, 2333333
    else:
        spec, cadus, plain, syms = util.npp_case(nframes=50); sps = 2.0; bw = 0.002; order = 4
        x, _ = synth.modulate(syms, spec); fs, sr = 30e6, 15e6
    a = P.block(0, [1e-2, 1, 1, 65536], x)
    f = P.block(1, [fs, sr, 0.5, 31], a)
    c = np.ascontiguousarray(P.block(2, [bw, order, 1.0], f))
    return c, sps
for name in sys.argv[1:] or ["goes"]:
    c, sps = case(name)
    n = len(c)
    nsym_all = int(n / sps) + 10
    tseq, omseq = track(c, 8, sps, 0, 1, 1, nsym_all)
    print(name, "samples", n, "symbols", len(tseq), "omega std rel", np.std(omseq[2000:]) / sps)
    rng = np.random.default_rng(1)
    starts = rng.integers(20000, n - 60000, 400)
    for (nfast, G, Gom, nslow, freeze) in [(0, 1, 1, 4131, 0), (0, 1, 1, 2000, 0), (0,1,1,1000,0), (256, 8, 1, 768, 0), (256, 8, 1, 768, 1), (256, 8, 8, 768, 0), (192, 12, 1, 512, 1), (128, 16, 1, 512, 1), (256, 8, 1, 512, 1), (384, 6, 1, 640, 1), (128, 8, 1, 384, 1), (256,4,1,768,1), (512,4,1,512,1)]:
        ds, doms = [], []
        for s in starts:
            t, om = track(c, int(s), sps, nfast, G, Gom, nfast + nslow, freeze)
            te = t[-1]
            j = np.searchsorted(tseq, te)
            j = min(max(j, 1), len(tseq) - 1)
            jj = j if abs(tseq[j] - te) < abs(tseq[j - 1] - te) else j - 1
            ds.append(te - tseq[jj]); doms.append((om[-1] - omseq[jj]) / sps)
        ds = np.abs(np.array(ds)); doms = np.abs(np.array(doms))
        print(f"  fast {nfast:4d} xG {G:4.0f} omG {Gom:3.0f} frz {freeze} slow {nslow:5d} -> W={int((nfast+nslow)*sps):6d} samp | |dt| med {np.median(ds):.4f} p90 {np.quantile(ds,.9):.4f} p99 {np.quantile(ds,.99):.4f} max {ds.max():.3f} fail>0.05: {np.mean(ds>0.05):.3f} | dom p99 {np.quantile(doms,.99):.2e}")
