"""Are the frames a stateful engine decodes around the wrap of the periodic recording the reference's? GPU: two passes over the
recording through one pair of handles; CPU: the reference on the recording twice in a row. (development probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from satdump_amd import capi, synth
from oracle import pyref
wlname = sys.argv[1] if len(sys.argv) > 1 else "metop_ahrpt"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2100
wl = bench.WORKLOADS[wlname]
rec = synth.Recording(synth.SynthSpec(**wl["spec"]), frames, blocks=1)
x = rec.synth_range(0, rec.n_samples, device="cuda")
n = x.numel()
dem = capi.PskDemod(capi.demod_cfg(**wl["demod"])); fec = capi.FecDecoder(capi.fec_cfg(**wl["fec"]))
d_soft = torch.empty(2 * n + 64, dtype=torch.int8, device="cuda"); d_cadu = torch.empty((frames + 64, 1024), dtype=torch.uint8, device="cuda")
outs, softs = [], []
for p in range(3):
    ns = dem.process_dev(x.data_ptr(), n, capi.FMT_CF32, d_soft.data_ptr(), 2 * n + 64)
    nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), frames + 64)
    outs.append(d_cadu[:nf].cpu().numpy().copy()); softs.append(d_soft[:ns].cpu().numpy().copy())
got = np.concatenate(outs); gsoft = np.concatenate(softs)
orc = pyref.best()
xh = x.cpu().numpy()
r, refc, _, _ = bench.ref_decode(orc, wl, np.concatenate([xh, xh, xh]), want_syms=False)
m = min(len(got), len(refc))
bad = [i for i in range(m) if not np.array_equal(got[i], refc[i])]
print(wlname, frames, "gpu frames", len(got), "ref frames", len(refc), "differing frames at", bad[:12], "per pass", [len(o) for o in outs])
ms = min(len(gsoft), len(r["soft"]))
d = gsoft[:ms].astype(np.int32) - r["soft"][:ms].astype(np.int32)
w = np.flatnonzero(np.abs(d) > 8)
print("soft symbols", len(gsoft), len(r["soft"]), "beyond 8 LSB:", len(w), w[:10], "per-pass soft counts", [len(s) for s in softs])
