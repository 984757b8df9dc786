import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from satdump_amd import capi
n = 8_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = (torch.randn(2 * n, device="cuda", generator=g) * 0.3).contiguous()
for name, kw in [("goes", dict(samplerate=3e6, symbolrate=927000, constellation="bpsk", rrc_alpha=0.5, pll_bw=0.02, max_sps=3.0)),
                 ("npp", dict(samplerate=30e6, symbolrate=15e6, constellation="qpsk", rrc_alpha=0.5, pll_bw=0.002))]:
    dem = capi.PskDemod(capi.demod_cfg(**kw))
    d_soft = torch.zeros(2 * n + 64, dtype=torch.int8, device="cuda")
    for it in range(2):
        t0 = time.time()
        try:
            ns = dem.process_dev(x.data_ptr(), n, capi.FMT_CF32, d_soft.data_ptr(), 2 * n + 64)
            torch.cuda.synchronize()
            st = dem.stats()
            print(name, it, "ok", ns, f"{time.time()-t0:.3f}s chunks", st.chunks, "fixed", st.chunks_fixed)
        except Exception as e:
            print(name, it, "EXC", str(e)[:200], f"{time.time()-t0:.3f}s")
