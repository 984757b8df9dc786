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


def lockin_overlap(demod: dict, fec: dict) -> int:
    """Samples a rank reads in front of its own range so that a cold-started chain is producing the reference's frames when the
    range begins. Sum of the lock-in times of the stages, from the loop constants the modules are configured with:
      AGC 8/agc_rate samples, Costas 16/pll_bw samples, M&M 40/clock_gain_mu symbols; viterbi_outsync_after + 2 Viterbi blocks
      (the lock search runs on the first block, viterbi_1_2.cpp:52-92 -- and when it runs while the loops are still settling it
      can lock on a wrong phase with a BER just under the threshold, which the decoder only gives up after outsync_after bad
      blocks, :104-113; measured on a cold-started NPP chunk: first good frame 11 CADUs in with an unlucky start, 1 CADU in
      otherwise), for MetOp 10 more (the module's own no-sync watchdog, module_metop_ahrpt_decoder.cpp:58-66); and four CADUs
      (the deframer needs consecutive ASMs before it reports SYNCED, bpsk_ccsds_deframer.cpp:47-107; one more frame straddles the
      boundary)."""
    sps = float(demod["samplerate"]) / float(demod["symbolrate"])
    q = 1 if demod.get("constellation", "qpsk") == "bpsk" else 2
    metop = fec.get("decoder", 0) == 1
    cadu_bits = 8192 if metop else int(fec.get("cadu_size", 8192))
    conv_rate = 0.75 if metop else {0: 0.5, 1: 2 / 3, 2: 0.75, 3: 5 / 6, 4: 7 / 8}[int(fec.get("conv_rate", 0))]
    block_syms = (16384 if metop else max(cadu_bits, 8192)) / q
    cadu_syms = cadu_bits / conv_rate / q
    gmu = float(demod.get("clock_gain_mu", 8.7e-3))
    relock_blocks = int(fec.get("viterbi_outsync_after", 10 if metop else 20)) + 2 + (10 if metop else 0)
    n = 8.0 / float(demod.get("agc_rate", 1e-2)) + 16.0 / float(demod["pll_bw"]) + (40.0 / gmu + relock_blocks * block_syms + 4 * cadu_syms) * sps
    return int(n + 7) // 8 * 8


def overlap_drop(tail_prev: np.ndarray, head: np.ndarray) -> int:
    """Leading frames of `head` (first frames a rank decoded) that repeat the end of `tail_prev` (last frames of everything in
    front of it): the largest m with head[:m] == tail_prev[-m:]; if there is none, the first frames of head that occur anywhere
    in tail_prev (the rank's re-lock began in the middle of the overlap). Frames are compared behind their 4-byte sync marker:
    the marker is not RS protected, so two decodes of the same frame from differently started loops may differ in it, while the
    RS-corrected code block is the transmitted one in both. CADUs of a real recording carry counters, so genuine repeats do not occur."""
    tail_prev = np.asarray(tail_prev, dtype=np.uint8)
    head = np.asarray(head, dtype=np.uint8)
    if len(tail_prev) == 0 or len(head) == 0:
        return 0
    if tail_prev.shape[1] > 8:
        tail_prev, head = tail_prev[:, 4:], head[:, 4:]
    # candidates: positions of head[0] in tail_prev
    pos = np.flatnonzero((tail_prev == head[0][None, :]).all(axis=1))
    for p in pos:  # earliest position = largest overlap first
        m = len(tail_prev) - int(p)
        if m <= len(head) and np.array_equal(tail_prev[p:], head[:m]):
            return m
    seen = {bytes(r) for r in tail_prev}
    j = 0
    while j < len(head) and bytes(head[j]) in seen:
        j += 1
    return j


def edge_frames(overlap_samples: int, samples_per_frame: float, margin: int = 16) -> int:
    """Boundary frames a rank must contribute to stitch_plan: every frame that can lie inside the lock-in overlap, plus a margin."""
    return max(64, int(np.ceil(overlap_samples / max(1.0, samples_per_frame))) + margin)


def stitch_plan(heads, tails, counts, edge: int = 64):
    """Frames to drop at the head of every rank's CADU list, from the boundary frames alone: heads[r] / tails[r] = the first / last
    (up to `edge`) frames rank r decoded, counts[r] = how many it decoded. The running tail of the stitched stream is kept so that
    a rank that decoded fewer than `edge` frames does not hide its predecessor's.
    `edge` must cover the overlap (edge_frames()): two ranks that share MORE than `edge` frames cannot be told from two that share
    none by looking at `edge` boundary frames -- that case raises instead of emitting the shared frames twice."""
    drops = [0] * len(counts)
    run = np.zeros((0, 0), dtype=np.uint8)
    for r in range(len(counts)):
        h = np.asarray(heads[r], dtype=np.uint8)
        t = np.asarray(tails[r], dtype=np.uint8)
        c = int(counts[r])
        if c == 0:
            continue
        if run.size:
            drops[r] = overlap_drop(run, h)
            if len(h) >= edge and c > len(h):
                hk = h[:, 4:] if h.shape[1] > 8 else h
                rk = run[:, 4:] if run.shape[1] > 8 else run
                # every boundary frame of this rank repeats the predecessor, or the predecessor's oldest kept frame shows up inside
                # this rank's head: the overlap reaches beyond the `edge` frames that were exchanged
                if drops[r] >= len(h) or (drops[r] == 0 and len(rk) and (hk == rk[0][None, :]).all(axis=1).any()):
                    raise ValueError(f"rank {r}: the overlap with its predecessor exceeds the {edge} boundary frames exchanged (use edge_frames())")
        if c - drops[r] >= len(t) or not run.size:
            kept_tail = t if c - drops[r] >= len(t) else t[len(t) - (c - drops[r]):]
            run = kept_tail[-edge:] if c - drops[r] >= edge or not run.size else np.concatenate([run, kept_tail], axis=0)[-edge:]
        else:
            run = np.concatenate([run, t[len(t) - (c - drops[r]):]], axis=0)[-edge:]
    return drops


def stitch_cadus(per_rank_frames):
    """Concatenate per-rank CADU arrays [n_r, cadu_bytes] in rank order, dropping the leading frames of rank r that repeat the
    tail of what is already stitched (a frame transmitted inside the overlap region is decoded by both neighbours)."""
    fr = [np.asarray(f, dtype=np.uint8) for f in per_rank_frames]
    for f in fr:
        if f.ndim != 2:
            raise ValueError("frames must be [n, cadu_bytes]")
    edge = max([64] + [len(f) for f in fr])  # the full lists are at hand: any overlap size is found
    drops = stitch_plan([f[:edge] for f in fr], [f[-edge:] if len(f) else f for f in fr], [len(f) for f in fr], edge)
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
