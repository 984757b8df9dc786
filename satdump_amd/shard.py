"""Stream-parallel sharding of the hot path across ranks (one process per GPU) -- Python face of the C ABI's sdhip_shard_* entry points
(include/sdhip.h; the logic is satdump_amd/csrc/shard.hip since round 4, this file moves arrays in and out of it) plus the one thing that needs
torch.distributed: the reduction of the timing / counters.

The path has no data-path collective: a recording is cut into contiguous per-rank chunks, every rank demodulates its chunk
independently (its loops start cold, so it reads `overlap` samples in front of its range), finds where its soft-symbol stream CONTINUES its
predecessor's (sdhip_shard_align: a few KB of boundary symbols exchanged), decodes from the single stream's Viterbi block grid, and the per-rank
CADU lists are concatenated in rank order, dropping the frames two neighbours both decoded -- compared whole, sync marker and RS parity included:
the stitched list is the single stream's."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi as _capi


class ShardRange(C.Structure):
    _fields_ = [("read_start", C.c_uint64), ("own_start", C.c_uint64), ("stop", C.c_uint64)]


def _lib():
    L = _capi.lib()
    if not getattr(L, "_shard_bound", False):
        L.sdhip_shard_overlap.restype = C.c_uint64
        L.sdhip_shard_overlap.argtypes = [C.POINTER(_capi.DemodCfg), C.POINTER(_capi.FecCfg)]
        L.sdhip_shard_lockin.argtypes = [C.POINTER(_capi.DemodCfg), C.POINTER(_capi.FecCfg), C.POINTER(C.c_uint64)]
        L.sdhip_shard_plan.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.POINTER(ShardRange)]
        L.sdhip_shard_align.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.sdhip_shard_stitch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_int, C.c_int,
                                         C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
        L._shard_bound = True
    return L


def plan_chunks(n_samples: int, world: int, overlap: int, align: int = 8):
    """Rank r owns samples [r*n/world, (r+1)*n/world); it READS from `overlap` samples earlier (clipped at 0). Boundaries are multiples of `align`."""
    if world < 1:
        raise ValueError("world must be >= 1")
    out = (ShardRange * world)()
    if _lib().sdhip_shard_plan(int(n_samples), int(world), int(overlap), int(align), out) != 0:
        raise ValueError(_capi.last_error())
    return [{"rank": r, "read_start": int(o.read_start), "own_start": int(o.own_start), "stop": int(o.stop)} for r, o in enumerate(out)]


def lockin_overlap(demod: dict, fec: dict) -> int:
    """Samples a rank reads in front of its own range (sdhip_shard_overlap: the stages' lock-in times from the loop constants the modules are configured with)."""
    d = _capi.demod_cfg(**{k: v for k, v in demod.items() if k != "device"})
    f = _capi.fec_cfg(**{k: v for k, v in fec.items() if k != "device"})
    return int(_lib().sdhip_shard_overlap(C.byref(d), C.byref(f)))


def lockin_parts(demod: dict, fec: dict):
    """(samples until a cold-started demodulator is locked, soft bytes a cold-started decoder needs before the first frame that counts, decoder block bytes)."""
    d = _capi.demod_cfg(**{k: v for k, v in demod.items() if k != "device"})
    f = _capi.fec_cfg(**{k: v for k, v in fec.items() if k != "device"})
    out = (C.c_uint64 * 3)()
    if _lib().sdhip_shard_lockin(C.byref(d), C.byref(f), out) != 0:
        raise ValueError(_capi.last_error())
    return int(out[0]), int(out[1]), int(out[2])


def align_ranks(soft_tail_of, soft_head_of, n_soft: int, q: int, plan_me: dict, rank: int, world: int, all_gather, window_syms: int = 2048, radius: int = 8192):
    """The boundary exchange of the N-rank flow. soft_tail_of(nbytes) / soft_head_of(nbytes) fetch the last / first bytes of this rank's soft stream (host
    int8 arrays); all_gather(np.ndarray) -> list of every rank's array. Returns (lag in symbols: where this rank's stream continues its predecessor's, quarter
    turns against the predecessor, agreement, global symbol index of that continuation point, found)."""
    T = min(window_syms, max(64, n_soft // q // 4))
    tail = np.zeros(window_syms * q, dtype=np.int8)
    t = soft_tail_of(T * q)
    tail[window_syms * q - len(t):] = t
    tails = all_gather(tail)
    lag, turn, agree, found = 0, 0, 1.0, True
    if rank > 0:
        nsym = n_soft // q
        span = plan_me["stop"] - plan_me["read_start"]
        expect = int(nsym * (plan_me["own_start"] - plan_me["read_start"]) / max(1, span))
        head = soft_head_of(min(n_soft, (expect + radius + T) * q))
        lag, turn, agree, found = align(tails[rank - 1][(window_syms - T) * q:], head, q, expect, radius)
    own = np.array([n_soft // q - lag, lag, turn, int(found)], dtype=np.int64)
    owns = all_gather(own)
    before = int(sum(int(o[0]) for o in owns[:rank]))
    return lag, turn, agree, before, found


def align(prev_tail: np.ndarray, head: np.ndarray, q: int, expect: int = 0, radius: int = 0):
    """Where `head` (this chunk's first soft bytes) continues `prev_tail` (the predecessor's last soft bytes): (lag in symbols = index in head of the symbol that
    follows the predecessor's last one, quarter turns of this chunk's constellation against the predecessor's, agreement of the hard decisions, found)."""
    a = np.ascontiguousarray(prev_tail, dtype=np.int8)
    b = np.ascontiguousarray(head, dtype=np.int8)
    lag, turn, agree = C.c_int64(-1), C.c_int(0), C.c_float(0.0)
    rc = _lib().sdhip_shard_align(a.ctypes.data_as(C.c_void_p), a.size, b.ctypes.data_as(C.c_void_p), b.size, int(q), int(expect), int(radius), C.byref(lag), C.byref(turn),
                                  C.byref(agree))
    if rc < 0:
        raise ValueError(_capi.last_error())
    return int(lag.value), int(turn.value), float(agree.value), rc == 0


def fec_start(global_syms_before: int, lag: int, q: int, block_bytes: int, lockin_bytes: int, min_local_bytes: int = 0) -> int:
    """The byte of a chunk's soft stream its decoder starts at: the chunk's symbol `lag` is the stream's symbol `global_syms_before`; the decoder has to start
    on a multiple of block_bytes of the GLOBAL soft stream (the single stream's Viterbi buffers), at least lockin_bytes in front of the chunk's own first
    byte (its lock search, watchdog and deframer settle there) and not before min_local_bytes (where the chunk's demodulator has locked)."""
    own_global = global_syms_before * q
    start_global = max(0, (own_global - lockin_bytes)) // block_bytes * block_bytes
    local = lag * q - (own_global - start_global)
    while local < min_local_bytes:
        local += block_bytes
    return int(local)


def edge_frames(overlap_samples: int, samples_per_frame: float, margin: int = 16) -> int:
    """Boundary frames a rank must contribute to stitch_plan: every frame that can lie inside the lock-in overlap, plus a margin."""
    return max(64, int(np.ceil(overlap_samples / max(1.0, samples_per_frame))) + margin)


def stitch_plan(heads, tails, counts, edge: int = 64, whole_frames: bool = False):
    """Frames to drop at the head of every rank's CADU list, from the boundary frames alone (sdhip_shard_stitch): heads[r] / tails[r] = the first / last (up to
    `edge`) frames rank r decoded, counts[r] = how many it decoded. Raises when two ranks share more than `edge` frames."""
    world = len(counts)
    hs = [np.ascontiguousarray(h, dtype=np.uint8) for h in heads]
    ts = [np.ascontiguousarray(t, dtype=np.uint8) for t in tails]
    fb = next((a.shape[1] for a in hs + ts if a.ndim == 2 and a.shape[0]), 0)
    if fb == 0:
        return [0] * world
    hp = (C.c_void_p * world)(*[a.ctypes.data if a.size else None for a in hs])
    tp = (C.c_void_p * world)(*[a.ctypes.data if a.size else None for a in ts])
    nh = (C.c_size_t * world)(*[len(a) for a in hs])
    nt = (C.c_size_t * world)(*[len(a) for a in ts])
    cn = (C.c_uint64 * world)(*[int(c) for c in counts])
    drops = (C.c_uint64 * world)()
    if _lib().sdhip_shard_stitch(hp, nh, tp, nt, cn, world, int(fb), int(edge), int(bool(whole_frames)), drops) != 0:
        raise ValueError(_capi.last_error())
    return [int(d) for d in drops]


def stitch_cadus(per_rank_frames, whole_frames: bool = False):
    """Concatenate per-rank CADU arrays [n_r, cadu_bytes] in rank order, dropping the leading frames of rank r that repeat the tail of what is already
    stitched (a frame transmitted inside the overlap region is decoded by both neighbours)."""
    fr = [np.asarray(f, dtype=np.uint8) for f in per_rank_frames]
    for f in fr:
        if f.ndim != 2:
            raise ValueError("frames must be [n, cadu_bytes]")
    edge = max([64] + [len(f) for f in fr])  # the full lists are at hand: any overlap size is found
    drops = stitch_plan([f[:edge] for f in fr], [f[-edge:] if len(f) else f for f in fr], [len(f) for f in fr], edge, whole_frames)
    parts = [f[d:] for f, d in zip(fr, drops) if len(f) - d > 0]
    if not parts:
        return np.zeros((0, 0), dtype=np.uint8)
    return np.concatenate(parts, axis=0)


def reduce_metrics(dt_s: float, samples: float, frames: float, device=None):
    """(max over ranks of dt, sum of samples, sum of frames). With no initialised process group: identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(dt_s), float(samples), float(frames)
    t = torch.tensor([dt_s], dtype=torch.float64, device=device)
    c = torch.tensor([samples, frames], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c[0].item()), float(c[1].item())
