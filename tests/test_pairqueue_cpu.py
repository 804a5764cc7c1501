"""Multi-GPU pair queue logic on CPU: world_size-2 gloo processes (the N>1 path of SURVEY.md §8e)."""
import importlib
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    manifest = [{"seed": 100 + i} for i in range(7)] if rank == 0 else []
    seen = []

    def register(i, item):  # stands in for ctx.register_pair: a pure function of the manifest entry
        seen.append(i)
        return {"pair": i, "rank": rank, "value": item["seed"] * 2}

    out = pq.run_sharded(manifest, register, dist)
    q.put((rank, seen, out))
    dist.destroy_process_group()


def test_static_partition():
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    for n, w in ((64, 8), (7, 2), (3, 4), (0, 2)):
        parts = [pq.pairs_for_rank(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        pq.pairs_for_rank(4, 2, 2)
    assert pq.run_sharded([1, 2, 3], lambda i, x: {"v": x * x}) == [{"v": 1}, {"v": 4}, {"v": 9}]


def test_two_rank_gloo_pair_queue():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]  # pair p -> rank p mod 2
    for _, _, out in got:  # every rank holds every pair's record, in pair order
        assert [r["pair"] for r in out] == list(range(7))
        assert [r["rank"] for r in out] == [0, 1, 0, 1, 0, 1, 0]
        assert [r["value"] for r in out] == [2 * (100 + i) for i in range(7)]


def _mv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    n = 5
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j] if rank == 0 else []  # all ordered pairs of 5 clouds
    built = []

    def make_cloud(c):  # stands in for ctx.cloud_create
        built.append(c)
        return {"cloud": c, "rank": rank}

    def register_batch(ids, S, T):  # stands in for ctx.register_clouds
        assert all(h["rank"] == rank for h in S + T)
        return [{"pair": i, "s": s["cloud"], "t": t["cloud"], "rank": rank} for i, s, t in zip(ids, S, T)]

    out = pq.run_multiview(n, pairs, make_cloud, register_batch, dist)
    q.put((rank, built, out))
    dist.destroy_process_group()


def test_multiview_single_process_and_validation():
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    built = []
    pairs = [(0, 1), (2, 1), (1, 0), (0, 2), (0, 1)]
    out = pq.run_multiview(3, pairs, lambda c: built.append(c) or ("h", c), lambda ids, S, T: [{"p": i, "s": s[1], "t": t[1]} for i, s, t in zip(ids, S, T)])
    assert built == [0, 1, 2]  # every cloud's front end exactly once, whatever the number of pairs it takes part in
    assert [(r["s"], r["t"]) for r in out] == pairs and [r["p"] for r in out] == list(range(5))
    assert pq.clouds_for_pairs(pairs, [1]) == [1, 2]
    assert pq.run_multiview(3, [], lambda c: None, lambda ids, S, T: []) == []
    with pytest.raises(ValueError):
        pq.run_multiview(2, [(0, 2)], lambda c: c, lambda ids, S, T: [{}])
    with pytest.raises(RuntimeError):
        pq.run_multiview(2, [(0, 1)], lambda c: c, lambda ids, S, T: [])


def test_two_rank_gloo_multiview():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mv_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    pairs = [(i, j) for i in range(5) for j in range(5) if i != j]
    for rank, built, out in got:
        assert len(built) == len(set(built)) and set(built) <= set(range(5))  # a cloud is never built twice on a rank
        assert [(r["s"], r["t"]) for r in out] == pairs  # every rank ends up with every record, in pair order
        assert [r["rank"] for r in out] == [i % 2 for i in range(len(pairs))]  # pair p was registered on rank p mod 2
    assert got[0][2] == got[1][2]


def _records_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    # what bench.py does per job / per step, on CPU tensors: manifest from rank 0, static shard, one all-gather of the records
    manifest = pq.broadcast_manifest(pq.job_manifest(5, 3, world, False) if rank == 0 else [], dist)
    mine = pq.pairs_for_rank(len(manifest), rank, world)
    rows = pq.records_per_rank(len(manifest), world)
    results = [(10 + p, p % 2, [float(manifest[p] * 100 + k) for k in range(16)]) for p in mine]
    block = torch.from_numpy(pq.pack_records(mine, results, rows))
    out = pq.gather_records(block, dist)
    q.put((rank, manifest, mine, sorted(out.items())))
    dist.destroy_process_group()


def test_job_manifest_and_record_gather_two_ranks():
    """bench.py's job plumbing (manifest broadcast, static shard, per-step all-gather of the 19-double result records) with two gloo
    ranks: every rank ends up with the record of every pair of the job."""
    import torch.multiprocessing as mp

    pq = importlib.import_module("gh-icp_amd.pairqueue")
    assert pq.job_manifest(6, 4, 1, False) == [0, 1, 2, 3, 0, 1]
    assert pq.job_manifest(6, 4, 2, False) == [p % 8 for p in range(12)]     # weak: per-rank pairs and scenes
    assert pq.job_manifest(64, 64, 8, True) == list(range(64))              # strong: the fixed 64-pair job of cfg4
    assert pq.records_per_rank(7, 2) == 4 and pq.records_per_rank(0, 2) == 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_records_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    (r0, man0, mine0, out0), (r1, man1, mine1, out1) = got
    assert man0 == man1 == [p % 6 for p in range(10)]
    assert mine0 == [0, 2, 4, 6, 8] and mine1 == [1, 3, 5, 7, 9]
    assert out0 == out1 and [k for k, _ in out0] == list(range(10))
    for pid, (it, conv, Rt) in out0:
        assert it == 10 + pid and conv == pid % 2 and Rt == [float(man0[pid] * 100 + k) for k in range(16)]


def _dyn_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    # skewed job: the iterations of the bench scenes span 24 ... 112; here a pair "costs" its iteration count in milliseconds and the
    # slow pairs all sit on the even ids, i.e. on rank 0 under the static p mod R split
    iters = [112 if p % 2 == 0 and p < 24 else 24 for p in range(48)]
    manifest = [{"iters": it} for it in iters] if rank == 0 else []
    busy = {"static": 0.0, "dynamic": 0.0}

    def cost(item):
        time.sleep(item["iters"] * 1e-3)
        return item["iters"]

    def reg_one(i, item):
        t = time.perf_counter()
        v = cost(item)
        busy["static"] += time.perf_counter() - t
        return {"pair": i, "rank": rank, "iters": v}

    def reg_chunk(ids, items):
        t = time.perf_counter()
        out = [{"pair": i, "rank": rank, "iters": cost(it)} for i, it in zip(ids, items)]
        busy["dynamic"] += time.perf_counter() - t
        return out

    st = pq.run_sharded(manifest, reg_one, dist)
    dy = pq.run_sharded_dynamic(manifest, reg_chunk, dist, chunk=2, name="job_a")
    dy2 = pq.run_sharded_dynamic(manifest, reg_chunk, dist, name="job_b")  # default chunk, a second job on the same group
    q.put((rank, busy, st, dy, dy2))
    dist.destroy_process_group()


def test_dynamic_pair_queue_two_ranks_skewed():
    """The dynamic counter: every pair registered exactly once, every rank ends up with every record, and on a skewed job the busier
    rank carries far less than under the static split (max / mean of the per-rank busy time)."""
    import torch.multiprocessing as mp

    pq = importlib.import_module("gh-icp_amd.pairqueue")
    c = pq.SharedCounter()
    assert c.claim(3, 7) == [0, 1, 2] and c.claim(3, 7) == [3, 4, 5] and c.claim(3, 7) == [6] and c.claim(3, 7) == [] and c.claim(0, 7) == []
    assert pq.chunk_size(5376 * 8, 8) == 672 and pq.chunk_size(3, 8) == 1
    seen = []
    out = pq.run_sharded_dynamic([10, 20, 30], lambda ids, items: [seen.append(i) or {"v": x + 1} for i, x in zip(ids, items)])
    assert seen == [0, 1, 2] and out == [{"v": 11}, {"v": 21}, {"v": 31}]
    with pytest.raises(RuntimeError):
        pq.run_sharded_dynamic([1, 2], lambda ids, items: [])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dyn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    iters = [112 if p % 2 == 0 and p < 24 else 24 for p in range(48)]
    for _, _, st, dy, dy2 in got:
        for recs in (st, dy, dy2):
            assert [r["pair"] for r in recs] == list(range(48)) and [r["iters"] for r in recs] == iters
    assert got[0][3] == got[1][3] and got[0][4] == got[1][4]
    assert {r["rank"] for r in got[0][3]} == {0, 1}  # both ranks took part
    imb = {k: max(g[1][k] for g in got) / (sum(g[1][k] for g in got) / 2) for k in ("static", "dynamic")}
    assert imb["static"] > 1.3 and imb["dynamic"] < 1.15 and imb["dynamic"] < imb["static"], imb


def _one_rank_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    m = pq.broadcast_manifest([5, 3, 9], dist)
    blk = torch.from_numpy(pq.pack_records([2, 0], [(7, 1, list(range(16))), (9, 0, [0.5] * 16)], 3))
    out = pq.gather_records(blk, dist)  # a one-rank group still goes through all_gather (the RCCL self-test of tests/test_gpu_multirank.py)
    c = pq.SharedCounter(dist, "selftest")
    ids = c.claim(4, 6) + c.claim(4, 6) + c.claim(4, 6)
    q.put((m, sorted(out), out[2][0], out[0][2][0], ids, c._store is not None))
    dist.destroy_process_group()


def test_one_rank_group_runs_the_same_collectives():
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=120)
    p.join(30)
    assert got == ([5, 3, 9], [0, 2], 7, 0.5, [0, 1, 2, 3, 4, 5], True)


# ------------------------------------------------------------------ the queue behind the C ABI (ghicp_pairqueue_*, host transport: no GPU)
def _native_rank(args):
    """One rank of a ghicp_pairqueue job over the rendezvous segment (spawned process)."""
    path, rank, world = args
    import importlib
    import os
    import sys

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    q = pq.NativeQueue(path, rank, world, pq.PQ_HOST, None, timeout_s=60.0)
    out = {"rank": rank}
    # manifest: rank 0's list on every rank; a payload larger than the 4 MB data window goes through in parts
    out["manifest"] = q.broadcast_manifest([7, 3, 9, 11, 2] if rank == 0 else [])
    big = np.arange(1_300_000, dtype=np.int64).tobytes() if rank == 1 % world else b""
    got = q.broadcast_bytes(big, root=1 % world)
    out["big_ok"] = bool(np.array_equal(np.frombuffer(got, np.int64), np.arange(1_300_000)))
    # static split
    out["static"] = q.static_share(23)
    # dynamic split: two jobs on the same queue (counter_reset between them), claims of 3 below 40
    claims = []
    for job in range(2):
        q.counter_reset()
        mine = []
        while True:
            ids = q.claim(3, 40)
            if not ids:
                break
            mine += ids
        claims.append(mine)
    out["claims"] = claims
    # result records: this rank's pairs of the static split, one all-gather
    mine = out["static"]
    rows = pq.records_per_rank(23, world)
    block = pq.pack_records(mine, [(10 + p, p % 2, [p + 0.5 * k for k in range(16)]) for p in mine], rows)
    out["records"] = {k: v for k, v in q.gather_records(block).items()}
    # a block larger than the window's per-rank share (rows x 152 B > 4 MB / world)
    nbig = 40_000
    blk = np.zeros((nbig, pq.RECORD_WIDTH))
    blk[:, 0] = np.arange(nbig) * world + rank
    blk[:, 5] = rank + 1
    allr = q.gather_records(blk)
    out["big_gather_ok"] = len(allr) == nbig * world and all(allr[p * world + r][2][2] == r + 1 for p in (0, 17, nbig - 1) for r in range(world))
    q.barrier()
    q.close()
    return out


@pytest.mark.parametrize("world", [1, 2, 3])
def test_native_pair_queue_host_transport(world, tmp_path):
    """ghicp_pairqueue_* (include/ghicp_c.h) with `world` processes over the rendezvous segment: the manifest broadcast, the static
    p mod R split, disjoint and complete claims on the shared counter over two jobs, and the all-gather of the 19-double records --
    the operations the RCCL transport runs as ncclBroadcast / ncclAllGather (tests/test_gpu_multirank.py runs those on the MI355X)."""
    import multiprocessing as mp

    path = str(tmp_path / "pq_rendezvous")
    with mp.get_context("spawn").Pool(world) as pool:
        res = pool.map(_native_rank, [(path, r, world) for r in range(world)], chunksize=1)
    assert not os.path.exists(path)  # rank 0 removed the segment
    for r in res:
        assert r["manifest"] == [7, 3, 9, 11, 2] and r["big_ok"] and r["big_gather_ok"]
        assert r["static"] == list(range(r["rank"], 23, world))
        assert sorted(r["records"]) == list(range(23))
        for p, (it, conv, Rt) in r["records"].items():
            assert (it, conv) == (10 + p, p % 2) and Rt == [p + 0.5 * k for k in range(16)]
    for job in range(2):
        claimed = sorted(i for r in res for i in r["claims"][job])
        assert claimed == list(range(40))  # every pair exactly once, whoever drew it
        for r in res:  # chunks of consecutive ids
            c = r["claims"][job]
            assert all(c[i + 1] == c[i] + 1 or (c[i] + 1) % 3 == 0 or c[i] == 39 for i in range(len(c) - 1))


def test_native_pair_queue_argument_errors(tmp_path):
    import ctypes as C
    import importlib

    api = importlib.import_module("gh-icp_amd.api")
    lib = api.load()
    h = C.c_void_p()
    # RCCL needs a context (its device and stream); rank outside the world; a rank that never finds rank 0's segment times out
    assert lib.ghicp_pairqueue_create(None, str(tmp_path / "a").encode(), 0, 1, 1, C.c_double(1.0), C.byref(h)) == 1
    assert lib.ghicp_pairqueue_create(None, str(tmp_path / "a").encode(), 2, 2, 0, C.c_double(1.0), C.byref(h)) == 1
    assert lib.ghicp_pairqueue_create(None, str(tmp_path / "never").encode(), 1, 2, 0, C.c_double(0.5), C.byref(h)) == 5
    assert lib.ghicp_pairqueue_create(None, str(tmp_path / "one").encode(), 0, 1, 0, C.c_double(1.0), C.byref(h)) == 0
    first, n = C.c_int64(-1), C.c_int64(-1)
    assert lib.ghicp_pairqueue_claim(h, C.c_int64(5), C.c_int64(7), C.byref(first), C.byref(n)) == 0 and (first.value, n.value) == (0, 5)
    assert lib.ghicp_pairqueue_claim(h, C.c_int64(5), C.c_int64(7), C.byref(first), C.byref(n)) == 0 and (first.value, n.value) == (5, 2)
    assert lib.ghicp_pairqueue_claim(h, C.c_int64(5), C.c_int64(7), C.byref(first), C.byref(n)) == 0 and (first.value, n.value) == (7, 0)
    assert lib.ghicp_pairqueue_broadcast(h, None, C.c_int64(8), 0) == 1 and b"bad argument" in lib.ghicp_pairqueue_last_error(h)
    assert lib.ghicp_pairqueue_destroy(h) == 0
