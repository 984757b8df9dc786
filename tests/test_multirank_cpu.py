"""N>1 path on CPU: world_size-2 gloo processes run the sharding plan of satdump_amd/shard.py on ONE synthetic recording,
each rank decoding its chunk with the ORACLE standing in for the GPU engines (test infrastructure: there is no GPU here
and the product has no CPU path), rank 0 gathers and stitches. The stitched CADU list must equal what the sequential
reference decodes from the whole recording (minus the frames lost while the very first rank locks, which the sequential
run loses too), and the timing/counter reduction must be max / sum over ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from satdump_amd import shard  # noqa: E402


def test_plan_covers_the_recording_once():
    plan = shard.plan_chunks(1_000_003, 4, overlap=50_000)
    assert plan[0]["read_start"] == 0 and plan[0]["own_start"] == 0 and plan[-1]["stop"] == 1_000_003
    for a, b in zip(plan[:-1], plan[1:]):
        assert a["stop"] == b["own_start"] and b["own_start"] % 8 == 0
        assert b["read_start"] == b["own_start"] - 50_000


def test_stitch_drops_frames_decoded_twice():
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(20, 64), dtype=np.uint8)
    a, b, c = frames[:9], frames[6:15], frames[15:]
    out = shard.stitch_cadus([a, b, c])
    assert np.array_equal(out, frames)
    assert len(shard.stitch_cadus([a, np.zeros((0, 64), np.uint8), frames[9:]])) == 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, x_path, n, overlap, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyref
    x = np.load(x_path, mmap_mode="r")
    me = shard.plan_chunks(n, world, overlap)[rank]
    chunk = np.ascontiguousarray(x[me["read_start"]:me["stop"]])
    orc = pyref.best()
    soft = orc.psk_demod(pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0), chunk,
                         want_syms=False)["soft"]
    cadu = orc.concat_decode(pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=1), soft)["cadu"]
    gathered = [None] * world
    dist.all_gather_object(gathered, cadu)
    dt, ns, nf = shard.reduce_metrics(1.0 + rank, float(me["stop"] - me["own_start"]), float(len(cadu)))
    if rank == 0:
        np.save(os.path.join(out_dir, "stitched.npy"), shard.stitch_cadus(gathered))
        np.save(os.path.join(out_dir, "metrics.npy"), np.array([dt, ns, nf, sum(len(g) for g in gathered)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_one_recording(tmp_path):
    from oracle import pyref
    from satdump_amd import synth
    from tests import util
    spec, cadus, plain, syms = util.goes_case(nframes=60)
    x, _ = synth.modulate(syms, spec)
    n = len(x)
    x_path = str(tmp_path / "x.npy")
    np.save(x_path, x)
    orc = pyref.best()
    soft = orc.psk_demod(pyref.demod_cfg(samplerate=3e6, symbolrate=927000, constellation=pyref.BPSK, pll_bw=0.02, max_sps=3.0), x, want_syms=False)["soft"]
    want = orc.concat_decode(pyref.fec_cfg(constellation=pyref.BPSK, nrzm=1, rs_usecheck=1), soft)["cadu"]
    world = 2
    overlap = 6 * 53023  # six CADUs of lead-in: loops + Viterbi + deframer (needs ~3 consecutive ASMs) lock before the owned range
    mp.spawn(_worker, args=(world, _free_port(), x_path, n, overlap, str(tmp_path)), nprocs=world, join=True)
    got = np.load(str(tmp_path / "stitched.npy"))
    dt, ns, nf, ntot = np.load(str(tmp_path / "metrics.npy"))
    assert dt == 2.0 and ns == float(n) and nf == ntot  # max over ranks / sums over ranks
    # identical frame list (the sharded run may only differ by frames around the chunk boundary that BOTH decoded: stitched away)
    assert got.shape == want.shape and np.array_equal(got, want)
    ids = util.frame_ids(got, plain)
    assert all(i >= 0 for i in ids[2:]) and len(set(ids[2:])) == len(ids[2:])
