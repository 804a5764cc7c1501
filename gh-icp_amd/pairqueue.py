"""Pair queue for multi-GPU registration of independent scan pairs (BASELINE configs[3]).

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests).
Pairs share nothing, so the data path has NO collective: rank r takes pairs r, r+R, r+2R ... (static, deterministic
-> testable on one GPU with R virtual ranks).  The only communication is the pair manifest (broadcast from rank 0)
and the small per-pair result records (all-gather): a few KB in total.
"""
from __future__ import annotations

from typing import Callable, List, Sequence


def pairs_for_rank(n_pairs: int, rank: int, world: int) -> List[int]:
    """Static round-robin partition: pair p -> rank p mod world."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_pairs, world))


def run_sharded(manifest: Sequence, register_fn: Callable[[int, object], dict], dist=None) -> List[dict]:
    """Every rank calls this with the same (or, on non-zero ranks, any) manifest.
    `register_fn(pair_id, item)` registers one pair on this rank's GPU and returns a small picklable record.
    Returns the records of ALL pairs in pair order on every rank."""
    if dist is None or not dist.is_initialized():
        return [register_fn(i, item) for i, item in enumerate(manifest)]
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [list(manifest) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)  # the manifest: seeds / paths / parameters, not point data
    manifest = box[0]
    mine = [(i, register_fn(i, manifest[i])) for i in pairs_for_rank(len(manifest), rank, world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    out = [None] * len(manifest)
    for part in gathered:
        for i, rec in part:
            out[i] = rec
    return out
