"""Per-clip data parallelism for the eval loop.

The reference is single-process (scripts/finetune/inference_hyper_lora.py:1466-1479).  Each clip's generate() is
independent, so clips are sharded across ranks (one process per GPU, full weight replica) and the only exchange is
a gather of fixed-size result records to rank 0: {clip_id:int64, ids:int64[n_new]} (+ optional first-step logits).
`torch.distributed` backend "nccl" is RCCL on ROCm (peer->root transfers ride direct xGMI links); "gloo" is used by
the CPU tests.  No collective sits on the data path between clips.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def shard_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Round-robin clip -> rank assignment (clip i -> rank i mod W), as SURVEY.md 8e."""
    return list(range(rank, n_clips, world))


def block_of(n_clips: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [first, first + count) of rank `rank` when n_clips are spread over `world` ranks as evenly as possible (the first
    n_clips % world ranks hold one more): the strong-scaling partition of bench.py --strong and of an eval set whose size the world does not divide."""
    base, extra = divmod(n_clips, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_results(ids: torch.Tensor, clip0: int, world: int, rank: int, logits: Optional[torch.Tensor] = None
                   ) -> Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]]:
    """ids [B, n_new] int64 of clips clip0..clip0+B-1 on this rank -> on rank 0: (clip_ids [sum B], ids [sum B, n_new],
    logits [sum B, V] or None) ordered by clip id; other ranks return None.  B may DIFFER between ranks (a clip count the world size does
    not divide; a rank may even hold zero clips): every rank pads its records to the largest B (one all_reduce(MAX) of a single word) with
    clip id -1 and rank 0 drops the padding, so the gather itself stays a fixed-size `dist.gather` per payload (RCCL: one peer->root
    transfer per rank over its own xGMI link).  The SEQUENCE of collectives is the same on every rank whatever it holds: one all_reduce
    reconciles {largest B, whether anyone carries logits, n_new, V}; a rank without clips may pass `logits=None` (or ids of width 0) and
    takes the widths of the others, a rank WITH clips whose n_new / V / logits-presence disagrees with another's raises on every rank alike
    (before the first gather, so nobody hangs in a mismatched collective)."""
    B = ids.shape[0]
    cid = torch.arange(clip0, clip0 + B, device=ids.device, dtype=torch.int64)
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized() and os.environ.get("CRAB_BENCH_FORCE_DIST") == "1"):
        return cid, ids, logits                            # (the hook runs the collective with one rank: a one-GPU rehearsal of the RCCL path)
    if dist.get_backend() == "gloo":                      # CPU tests / one-GPU rehearsals: gloo gathers host tensors
        ids, cid = ids.cpu(), cid.cpu()
        logits = logits.cpu() if logits is not None else None
    # one word vector, reduced with MAX: [B, has_logits, n_new, V, -n_new, -V, -has_logits] - the negated entries give the MIN over the ranks
    # that hold clips (a rank without clips contributes the identity of both)
    BIG = 1 << 40
    has = 1 if logits is not None else 0
    nn_, vv = int(ids.shape[1]), (int(logits.shape[1]) if logits is not None else 0)
    if B > 0:
        words = [B, has, nn_, vv, -nn_, (-vv if has else -BIG), -has]      # V only constrains ranks that carry logits
    else:
        words = [0, 0, 0, 0, -BIG, -BIG, -1]
    w = torch.tensor(words, device=ids.device, dtype=torch.int64)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    Bm, any_logits, n_max, v_max, n_min, v_min, all_logits = int(w[0]), int(w[1]), int(w[2]), int(w[3]), -int(w[4]), -int(w[5]), -int(w[6])
    if Bm > 0 and (n_min != n_max or (any_logits and (all_logits != 1 or (v_min != v_max and v_min < BIG)))):
        raise ValueError(f"gather_results: ranks disagree on the record (n_new {n_min}..{n_max}, logits on some ranks only: {all_logits != any_logits}, "
                         f"V {v_min}..{v_max}): every rank that holds clips must pass the same n_new, and logits either everywhere or nowhere")
    if B == 0:                                            # nothing to send: take the record widths of the others
        ids = torch.empty((0, n_max), device=ids.device, dtype=torch.int64)
        logits = torch.empty((0, v_max), device=ids.device, dtype=torch.float32) if any_logits else None
    rec = torch.cat([cid[:, None], ids.to(torch.int64)], dim=1)
    if B < Bm:                                            # uneven shard: pad with clip id -1 (dropped on rank 0)
        rec = torch.cat([rec, torch.full((Bm - B, rec.shape[1]), -1, device=rec.device, dtype=torch.int64)], 0)
        if logits is not None:
            logits = torch.cat([logits, torch.zeros((Bm - B, logits.shape[1]), device=logits.device, dtype=logits.dtype)], 0)
    rec = rec.contiguous()
    bufs = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    dist.gather(rec, bufs, dst=0)
    lbufs = None
    if logits is not None:
        logits = logits.contiguous()
        lbufs = [torch.empty_like(logits) for _ in range(world)] if rank == 0 else None
        dist.gather(logits, lbufs, dst=0)
    if rank != 0:
        return None
    allrec = torch.cat(bufs, dim=0)
    live = allrec[:, 0] >= 0
    lg = torch.cat(lbufs, dim=0)[live] if lbufs is not None else None
    allrec = allrec[live]
    order = torch.argsort(allrec[:, 0])
    allrec = allrec[order]
    lg = lg[order] if lg is not None else None
    return allrec[:, 0], allrec[:, 1:], lg
