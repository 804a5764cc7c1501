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


# ----------------------------------------------------------------------------------------------------------------------
# Batched variant used by bench.py: the manifest is broadcast ONCE per job, every step ends with ONE all-gather of fixed-size
# result records (tensors on the ranks' device: RCCL on GPUs, gloo on CPU in the tests).
RECORD_WIDTH = 19  # pair id, iterations, converged, 16 entries of the 4x4 (row-major)


def job_manifest(pairs_per_step: int, distinct: int, world: int, strong: bool):
    """Scene id of every pair of ONE step of the whole job.  Weak scaling: `pairs_per_step` pairs and `distinct` scenes PER RANK;
    strong scaling (a fixed job, BASELINE cfg4): `pairs_per_step` pairs and `distinct` scenes in total."""
    n_job = pairs_per_step if strong else pairs_per_step * world
    n_scenes = max(1, min(n_job, distinct if strong else distinct * world))
    return [p % n_scenes for p in range(n_job)]


def broadcast_manifest(manifest, dist=None):
    """Rank 0's manifest on every rank (scene ids / seeds / paths: never point data)."""
    if dist is None or not dist.is_initialized():
        return list(manifest)
    box = [list(manifest) if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def records_per_rank(n_job: int, world: int) -> int:
    """Rows of the per-rank record block (equal on every rank: all_gather wants equal shapes; unused rows carry pair id -1)."""
    return max(1, -(-n_job // world))


def pack_records(mine: Sequence[int], results, rows: int):
    """`results[i]` (iterations, converged, Rt[16]) of pair `mine[i]` -> (rows, RECORD_WIDTH) float64 numpy block."""
    import numpy as np

    rec = np.zeros((rows, RECORD_WIDTH))
    rec[:, 0] = -1
    for i, (pid, r) in enumerate(zip(mine, results)):
        rec[i, 0], rec[i, 1], rec[i, 2] = pid, r[0], r[1]
        rec[i, 3:] = r[2]
    return rec


def gather_records(block, dist=None):
    """All ranks' record blocks (torch tensors of equal shape) -> dict pair id -> (iterations, converged, Rt16 list)."""
    blocks = [block]
    if dist is not None and dist.is_initialized():  # also for a one-rank group: the same collective runs whatever the world size
        import torch

        blocks = [torch.zeros_like(block) for _ in range(dist.get_world_size())]
        dist.all_gather(blocks, block)
    out = {}
    for b in blocks:
        for row in b.cpu().numpy():
            if row[0] >= 0:
                out[int(row[0])] = (int(row[1]), int(row[2]), [float(v) for v in row[3:]])
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Dynamic variant ("or a dynamic counter for skewed sizes").  The iterations a pair needs are not known in advance
# (24 ... 112 on the bench scenes), so the static p mod R split leaves ranks waiting for the rank that drew the slow pairs.  Here the
# ranks claim CHUNKS of consecutive pair ids from ONE shared counter -- an atomic add on torch.distributed's key-value store (the TCP
# store rank 0 owns; host side, a few bytes per claim) -- until the job is drained.  Still no collective on the data path: what is
# exchanged is the manifest, one integer per claim and the result records.
class SharedCounter:
    """Atomic counter shared by the ranks of a process group; a plain local counter without one."""

    def __init__(self, dist=None, name: str = "ghicp_pairqueue"):
        self._local = 0
        self._store = None
        self._key = name
        if dist is not None and dist.is_initialized():
            from torch.distributed import distributed_c10d as c10d

            self._store = c10d._get_default_store()

    def claim(self, count: int, limit: int) -> List[int]:
        """The next `count` ids below `limit` (fewer at the end, [] when the queue is drained)."""
        if count <= 0:
            return []
        if self._store is None:
            lo = self._local
            self._local += count
        else:
            lo = int(self._store.add(self._key, count)) - count  # add() returns the value AFTER the addition, atomically
        return list(range(min(lo, limit), min(lo + count, limit)))


def chunk_size(n_pairs: int, world: int, chunks_per_rank: int = 8) -> int:
    """Pairs per claim: small enough that the last chunks even out the ranks, large enough that a claim feeds a batched launch."""
    return max(1, n_pairs // max(1, world * chunks_per_rank))


def run_sharded_dynamic(manifest: Sequence, register_chunk: Callable[[List[int], List[object]], List[dict]], dist=None, chunk: int = 0,
                        name: str = "ghicp_pairqueue") -> List[dict]:
    """Like run_sharded, but the ranks claim chunks of pair ids from a SharedCounter until the job is drained.
    `register_chunk(pair_ids, items)` registers the chunk's pairs on this rank's GPU (one batched call) and returns one picklable
    record per pair.  `name` must be new for every job of a process group (the counter lives in the group's store).
    Returns the records of ALL pairs in pair order on every rank."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world > 1:
        box = [list(manifest) if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        manifest = box[0]
    n = len(manifest)
    counter = SharedCounter(dist, name)
    step = chunk or chunk_size(n, world)
    mine = []
    while True:
        ids = counter.claim(step, n)
        if not ids:
            break
        recs = register_chunk(ids, [manifest[i] for i in ids])
        if len(recs) != len(ids):
            raise RuntimeError("register_chunk returned %d records for %d pairs" % (len(recs), len(ids)))
        mine += list(zip(ids, recs))
    if world == 1:
        gathered = [mine]
    else:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    out = [None] * n
    for part in gathered:
        for i, rec in part:
            out[i] = rec
    return out


# ----------------------------------------------------------------------------------------------------------------------
# The same queue behind the C ABI (include/ghicp_c.h, "Pair queue"; gh-icp_amd/csrc/pairqueue.hip): what a C++ caller of the drop-in
# headers uses to shard pairs -- ncclBroadcast / ncclAllGather over RCCL, or the rendezvous segment alone (GHICP_PQ_HOST: ranks that
# share a GPU, machines without one).  This class is a ctypes binding over it, nothing more.
PQ_HOST, PQ_RCCL = 0, 1


class NativeQueue:
    """ghicp_pairqueue_*: create is collective (every rank, same `rendezvous` path, new per queue)."""

    def __init__(self, rendezvous: str, rank: int, world: int, transport: int = PQ_HOST, ctx=None, timeout_s: float = 120.0):
        import ctypes as C
        import importlib

        api = importlib.import_module("gh-icp_amd.api")
        self._C, self._api = C, api
        self.lib = ctx.lib if ctx is not None else api.load()
        self.lib.ghicp_pairqueue_last_error.restype = C.c_char_p
        self.ctx, self.rank, self.world, self.transport = ctx, int(rank), int(world), int(transport)
        self.h = C.c_void_p()
        rc = self.lib.ghicp_pairqueue_create(ctx.h if ctx is not None else None, rendezvous.encode(), C.c_int32(rank), C.c_int32(world),
                                             C.c_int32(transport), C.c_double(timeout_s), C.byref(self.h))
        if rc != 0:
            why = self.lib.ghicp_last_error(ctx.h).decode() if ctx is not None else ""
            raise api.GhicpError("ghicp_pairqueue_create failed (%d) %s" % (rc, why))

    def _check(self, rc):
        if rc != 0:
            raise self._api.GhicpError("pair queue error %d: %s" % (rc, self.lib.ghicp_pairqueue_last_error(self.h).decode()))

    def close(self):
        if self.h:
            h, self.h = self.h, None
            self._check(self.lib.ghicp_pairqueue_destroy(h))

    def broadcast_bytes(self, data: bytes, root: int = 0) -> bytes:
        """Root's bytes on every rank (two broadcasts: the length, then the payload)."""
        import numpy as np

        C = self._C
        n = np.array([len(data) if self.rank == root else 0], np.int64)
        self._check(self.lib.ghicp_pairqueue_broadcast(self.h, n.ctypes.data_as(C.c_void_p), C.c_int64(8), C.c_int32(root)))
        buf = np.frombuffer(data, np.uint8).copy() if self.rank == root else np.zeros(int(n[0]), np.uint8)
        self._check(self.lib.ghicp_pairqueue_broadcast(self.h, buf.ctypes.data_as(C.c_void_p), C.c_int64(int(n[0])), C.c_int32(root)))
        return buf.tobytes()

    def broadcast_manifest(self, manifest, root: int = 0):
        import json

        return json.loads(self.broadcast_bytes(json.dumps(list(manifest)).encode() if self.rank == root else b"", root).decode())

    def barrier(self):
        self._check(self.lib.ghicp_pairqueue_barrier(self.h))

    def static_share(self, n_pairs: int) -> List[int]:
        import numpy as np

        C = self._C
        ids = np.zeros(max(1, -(-n_pairs // self.world)), np.int64)
        n = C.c_int64(0)
        self._check(self.lib.ghicp_pairqueue_static_share(self.h, C.c_int64(n_pairs), ids.ctypes.data_as(C.c_void_p), C.c_int64(ids.size), C.byref(n)))
        return [int(v) for v in ids[:n.value]]

    def claim(self, count: int, limit: int) -> List[int]:
        C = self._C
        first, n = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.ghicp_pairqueue_claim(self.h, C.c_int64(count), C.c_int64(limit), C.byref(first), C.byref(n)))
        return list(range(first.value, first.value + n.value))

    def counter_reset(self):
        self._check(self.lib.ghicp_pairqueue_counter_reset(self.h))

    def gather_records(self, block):
        """block: (rows, RECORD_WIDTH) float64 numpy array of this rank -> dict pair id -> (iterations, converged, Rt16 list) of all ranks."""
        import numpy as np

        C = self._C
        block = np.ascontiguousarray(block, np.float64)
        assert block.ndim == 2 and block.shape[1] == RECORD_WIDTH
        out = np.zeros((self.world * block.shape[0], RECORD_WIDTH))
        self._check(self.lib.ghicp_pairqueue_gather_records(self.h, block.ctypes.data_as(C.c_void_p), C.c_int64(block.shape[0]), out.ctypes.data_as(C.c_void_p)))
        return {int(r[0]): (int(r[1]), int(r[2]), [float(v) for v in r[3:]]) for r in out if r[0] >= 0}

    def register_pairs(self, cfg, clouds_S, clouds_T, chunk: int = 0):
        """ghicp_pairqueue_register_pairs: device tensors (n_i, stride) f32 indexed by GLOBAL pair id (entries of pairs another rank registers
        may be None).  Returns the (n_pairs, RECORD_WIDTH) records of ALL pairs on every rank."""
        import numpy as np

        C = self._C
        n = len(clouds_S)
        stride = next(int(c.shape[1]) for c in list(clouds_S) + list(clouds_T) if c is not None)
        xs = (C.c_void_p * n)(*[c.data_ptr() if c is not None else None for c in clouds_S])
        xt = (C.c_void_p * n)(*[c.data_ptr() if c is not None else None for c in clouds_T])
        ns = (C.c_int64 * n)(*[int(c.shape[0]) if c is not None else 0 for c in clouds_S])
        nt = (C.c_int64 * n)(*[int(c.shape[0]) if c is not None else 0 for c in clouds_T])
        rec = np.zeros((n, RECORD_WIDTH))
        self._check(self.lib.ghicp_pairqueue_register_pairs(self.h, self.ctx.h, C.byref(cfg), C.c_int64(n), xs, ns, xt, nt, C.c_int(stride), C.c_int64(chunk),
                                                            None, rec.ctypes.data_as(C.c_void_p)))
        return rec
