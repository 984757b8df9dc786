"""How long does each feedback loop need to merge BITWISE with the sequential trajectory when restarted from a
default state mid-stream? (CPU study with the oracle blocks; calibrates the warm-up lengths.)"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import pyref
from satdump_amd import synth
from tests import util
P = pyref.port()
def merge_point(full, part, off_full):
    """first index i in part such that part[i:] == full[off_full+i:] bitwise (same alignment); -1 if never"""
    n = min(len(part), len(full) - off_full)
    eq = part[:n].view(np.uint64) == full[off_full:off_full + n].view(np.uint64)
    bad = np.flatnonzero(~eq)
    if len(bad) == 0: return 0
    if bad[-1] == n - 1: return -1
    return int(bad[-1] + 1)
for case in ["goes", "metop", "npp"]:
    if case == "goes":
        spec, cadus, plain, syms = util.goes_case(nframes=24); sps = 2.9126; bw = 0.02; order = 2
        x, _ = synth.modulate(syms, spec); x = P.block(4, [2700000, 3000000], x); fs, sr = 2.7e6, 927000
    elif case == "metop":
        spec, cadus, plain, syms = util.metop_case(nframes=40); sps = 2.5714290142; bw = 0.003; order = 4
        x, _ = synth.modulate(syms, spec); fs, sr = 6e6, 2333333
    else:
        spec, cadus, plain, syms = util.npp_case(nframes=40); sps = 2.0; bw = 0.002; order = 4
        x, _ = synth.modulate(syms, spec); fs, sr = 30e6, 15e6
    a = P.block(0, [1e-2, 1, 1, 65536], x)
    f = P.block(1, [fs, sr, 0.5, 31], a)
    c = P.block(2, [bw, order, 1.0], f)
    mmp = [sps, (8.7e-3) ** 2 / 4, 0.5, 8.7e-3, 0.005]
    m = P.block(3, mmp, c)
    rng = np.random.default_rng(0)
    starts = rng.integers(50000, len(x) - 120000, 24)
    # AGC merge (restart from gain 1)
    ag = [merge_point(a, P.block(0, [1e-2, 1, 1, 65536], x[s:s + 100000]), s) for s in starts]
    print(case, "AGC merge samples: med", int(np.median(ag)), "max", max(ag), "fails", sum(v < 0 for v in ag), "gain~", 1 / np.mean(np.abs(x)))
    # MM merge on the exact costas output: need symbol alignment -> search alignment by matching tail
    res = []
    for s in starts:
        part = P.block(3, mmp, c[s:s + 100000])
        # find alignment: match last symbol of part in m
        tail = part[-1].view(np.uint64) if False else None
        key = part[-1:].view(np.uint64)[0]
        idx = np.flatnonzero(m.view(np.uint64) == key)
        if len(idx) == 0: res.append(-1); continue
        off = idx[-1] - (len(part) - 1)
        mp = merge_point(m, part, off)
        res.append(mp)
    ok = [v for v in res if v >= 0]
    print(case, "MM merge symbols: med", int(np.median(ok)) if ok else None, "max", max(ok) if ok else None, "fails", sum(v < 0 for v in res), sorted(res))
    # Costas merge from (phase 0, freq 0): only bitwise-merge when landing in the same frame
    cs = []
    for s in starts:
        part = P.block(2, [bw, order, 1.0], f[s:s + 100000])
        cs.append(merge_point(c, part, s))
    ok = [v for v in cs if v >= 0]
    print(case, "Costas bitwise merges:", len(ok), "of", len(cs), "merge samples", sorted(ok))
