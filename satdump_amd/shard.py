"""Stream-parallel sharding of the hot path across ranks (one process per GPU) -- host logic only, no compute.

The path has no data-path collective: a recording is cut into contiguous per-rank chunks, every rank
demodulates and decodes its chunk independently (its loops restart from the reference's initial state, so each chunk
starts `overlap` samples early and the decoders re-lock inside that overlap), and the per-rank CADU lists are
concatenated in rank order on the host, dropping the frames two neighbouring ranks both decoded in the overlap.
The only torch.distributed traffic is the reduction of the timing / counters (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def plan_chunks(n_samples: int, world: int, overlap: int, align: int = 8):
    """Rank r owns samples [r*n/world, (r+1)*n/world); it READS from `overlap` samples earlier (clipped at 0) so that
    AGC / Costas / M&M / Viterbi / deframer are locked when its own range begins. Boundaries are multiples of `align`."""
    if world < 1:
        raise ValueError("world must be >= 1")
    edges = [(n_samples * r // world) // align * align for r in range(world)] + [n_samples]
    plan = []
    for r in range(world):
        own0, own1 = edges[r], edges[r + 1]
        plan.append({"rank": r, "read_start": max(0, own0 - overlap), "own_start": own0, "stop": own1})
    return plan


def stitch_cadus(per_rank_frames):
    """Concatenate per-rank CADU arrays [n_r, cadu_bytes] in rank order. A frame transmitted inside the overlap region may be
    decoded by both neighbours: the leading frames of rank r that repeat the tail of what is already stitched are dropped
    (byte comparison of whole frames; CADU payloads of real missions carry counters, so genuine repeats do not occur)."""
    out = None
    for f in per_rank_frames:
        f = np.asarray(f, dtype=np.uint8)
        if f.ndim != 2:
            raise ValueError("frames must be [n, cadu_bytes]")
        if out is None or len(out) == 0:
            out = f.copy()
            continue
        if len(f) == 0:
            continue
        # longest m such that f[:m] == out[-m:], searched from the largest plausible overlap down
        drop = 0
        max_m = min(len(f), len(out))
        tail_keys = [bytes(r) for r in out[-max_m:]]
        head_keys = [bytes(r) for r in f[:max_m]]
        for m in range(max_m, 0, -1):
            if tail_keys[-m:] == head_keys[:m]:
                drop = m
                break
        if drop == 0:
            # the overlap may also start in the MIDDLE of rank r's re-lock: find the first frame of f that continues `out`
            seen = {k: i for i, k in enumerate(tail_keys)}
            j = 0
            while j < len(head_keys) and head_keys[j] in seen:
                j += 1
            drop = j
        out = np.concatenate([out, f[drop:]], axis=0)
    return out if out is not None else np.zeros((0, 0), dtype=np.uint8)


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
