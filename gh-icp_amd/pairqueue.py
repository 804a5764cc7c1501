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


def clouds_for_pairs(pairs: Sequence[Sequence[int]], mine: Sequence[int]) -> List[int]:
    """Cloud ids (ascending) that the pairs with indices `mine` touch: the front ends one rank has to run."""
    need = set()
    for p in mine:
        s, t = pairs[p]
        need.add(int(s))
        need.add(int(t))
    return sorted(need)


def run_multiview(n_clouds: int, pairs: Sequence[Sequence[int]], make_cloud: Callable[[int], object],
                  register_batch: Callable[[List[int], List[object], List[object]], List[dict]], dist=None) -> List[dict]:
    """Multi-view registration of `pairs` = [(source cloud id, target cloud id), ...] over cached per-cloud front ends
    (ghicp_cloud_create / ghicp_register_clouds).  Rank r takes pairs r, r+R, ...; it builds the front
    end of every cloud its pairs touch exactly ONCE (`make_cloud(cloud_id)` -> handle) and registers all its pairs in one
    batched call (`register_batch(pair_ids, source_handles, target_handles)` -> one record per pair).  No data-path
    collective: the pair list is broadcast, the records are all-gathered.  Returns all records in pair order on every rank."""
    if dist is not None and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [[tuple(int(v) for v in p) for p in pairs] if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        pairs = box[0]
    else:
        rank, world = 0, 1
    for s, t in pairs:
        if not (0 <= s < n_clouds and 0 <= t < n_clouds):
            raise ValueError("pair refers to a cloud outside [0, %d)" % n_clouds)
    mine = pairs_for_rank(len(pairs), rank, world)
    handles = {c: make_cloud(c) for c in clouds_for_pairs(pairs, mine)}
    recs = register_batch(list(mine), [handles[pairs[p][0]] for p in mine], [handles[pairs[p][1]] for p in mine]) if mine else []
    if len(recs) != len(mine):
        raise RuntimeError("register_batch returned %d records for %d pairs" % (len(recs), len(mine)))
    part = list(zip(mine, recs))
    if world == 1:
        gathered = [part]
    else:
        gathered = [None] * world
        dist.all_gather_object(gathered, part)
    out = [None] * len(pairs)
    for g in gathered:
        for i, rec in g:
            out[i] = rec
    return out
