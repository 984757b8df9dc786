import sys; sys.path.insert(0, ".")
import numpy as np, torch, bench
from satdump_amd import capi
wl = bench.WORKLOADS["npp_hrd"]
frames = 32768
x, plain, spec = bench.make_input(wl, torch.device("cuda", 0), 0, frames)
n_in = x.numel()
dem = capi.PskDemod(capi.demod_cfg(**wl["demod"])); fec = capi.FecDecoder(capi.fec_cfg(**wl["fec"]))
d_soft = torch.empty(2 * n_in + 64, dtype=torch.int8, device="cuda"); d_cadu = torch.empty((frames + 64, 1024), dtype=torch.uint8, device="cuda")
for step in range(2):
    ns = dem.process_dev(x.data_ptr(), n_in, capi.FMT_CF32, d_soft.data_ptr(), 2 * n_in + 64)
    nf = fec.process_dev(d_soft.data_ptr(), ns, d_cadu.data_ptr(), frames + 64)
    ber, st = fec.block_taps()
    s = d_soft[:ns].cpu().numpy().astype(np.float32)
    blk = 8192
    pw = np.array([np.mean(np.abs(s[i*blk:(i+1)*blk])) for i in range(0, len(s)//blk, max(1, len(s)//blk//24))])
    print("step", step, "ns", ns, "nf", nf, "stats", fec.stats().tb_respec, fec.stats().vit_respec)
    print("  ber by 24ths:", np.round(ber[::max(1, len(ber)//24)], 3))
    print("  mean|soft| by 24ths:", np.round(pw, 1))
    dst = dem.stats(); print("  demod chunks", dst.chunks, "fixed", dst.chunks_fixed, "forced", dst.chunks_forced, "freq", dst.freq_hz)
