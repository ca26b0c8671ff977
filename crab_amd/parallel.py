"""Per-clip data parallelism for the eval loop.

The reference is single-process (scripts/finetune/inference_hyper_lora.py:1466-1479).  Each clip's generate() is
independent, so clips are sharded across ranks (one process per GPU, full weight replica) and the only exchange is
a gather of fixed-size result records to rank 0: {clip_id:int64, ids:int64[n_new]} (+ optional first-step logits).
`torch.distributed` backend "nccl" is RCCL on ROCm (peer->root transfers ride direct xGMI links); "gloo" is used by
the CPU tests.  No collective sits on the data path between clips.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def shard_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Round-robin clip -> rank assignment (clip i -> rank i mod W), as SURVEY.md 8e."""
    return list(range(rank, n_clips, world))


def gather_results(ids: torch.Tensor, clip0: int, world: int, rank: int, logits: Optional[torch.Tensor] = None
                   ) -> Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]]:
    """ids [B, n_new] int64 of clips clip0..clip0+B-1 on this rank -> on rank 0: (clip_ids [W*B], ids [W*B, n_new],
    logits [W*B, V] or None) ordered by clip id; other ranks return None.  Equal B on every rank (weak scaling)."""
    B = ids.shape[0]
    cid = torch.arange(clip0, clip0 + B, device=ids.device, dtype=torch.int64)
    if world == 1:
        return cid, ids, logits
    import torch.distributed as dist
    if dist.get_backend() == "gloo":                      # CPU tests / one-GPU rehearsals: gloo gathers host tensors
        ids, cid = ids.cpu(), cid.cpu()
        logits = logits.cpu() if logits is not None else None
    rec = torch.cat([cid[:, None], ids.to(torch.int64)], dim=1).contiguous()
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
    order = torch.argsort(allrec[:, 0])
    allrec = allrec[order]
    lg = torch.cat(lbufs, dim=0)[order] if lbufs is not None else None
    return allrec[:, 0], allrec[:, 1:], lg
