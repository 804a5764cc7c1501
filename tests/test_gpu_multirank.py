"""The N > 1 path of bench.py on REAL HIP contexts within one GPU ("on 1 GPU, validate the queue logic with R virtual
ranks"): two processes share device 0, the pair queue runs over gloo (RCCL refuses two ranks on one device; gloo does not), static and
dynamic; the all-gathered records of the job must be IDENTICAL to the one-rank run's -- same pairs, same iterations, same 4x4 bits."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp, ranks, extra):
    d = os.path.join(str(tmp), "r%d_%s" % (ranks, "_".join(a.strip("-") for a in extra if a.startswith("--"))))
    os.makedirs(d, exist_ok=True)
    args = ["--gpus", str(ranks), "--steps", "1", "--warmup", "1", "--cpu-baseline", "0", "--fe-batch", "8", "--fe-batch-streams", "2", "--detail-dir", d] + extra
    if ranks > 1:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args + ["--backend", "gloo"]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the JSON line"
    cfg = int(extra[extra.index("--config") + 1])
    detail = json.load(open(os.path.join(d, "bench_detail_cfg%d.json" % cfg)))
    return json.loads(lines[0]), detail["job_records"]


@pytest.mark.parametrize("queue", ["static", "dynamic"])
def test_two_virtual_ranks_on_one_gpu_equal_one_rank(ctx, tmp_path, queue):
    if getattr(ctx, "simulated", False):
        pytest.skip("bench.py on the interpreter with two ranks is tests/test_sim_cpu.py")
    # cfg4 is a FIXED job (strong scaling): 12 fragment pairs in all, whoever registers them
    job = ["--config", "4", "--hits", "40000", "--pairs-per-step", "12", "--distinct", "12"]
    one, rec1 = _run(tmp_path, 1, job)
    two, rec2 = _run(tmp_path, 2, job + ["--queue", queue, "--queue-chunks", "3"])
    assert two["n_gpus"] == 2 and len(two["rank_wall_s"]["per_rank"]) == 2 and two["config"]["backend"] == "gloo"
    assert ("(%s)" % queue) in two["config"]["parallelism"]
    assert sorted(rec1) == sorted(rec2) == sorted(str(i) for i in range(12))
    for k in rec1:
        assert rec1[k] == rec2[k], "pair %s differs between one rank and two" % k  # iterations, converged flag and the sixteen f64 of the 4x4


def test_two_virtual_ranks_weak_scaling_kuhn_munkres(ctx, tmp_path):
    """cfg2 (weak scaling, Kuhn-Munkres: the persistent pair loop): 2 ranks x 3 pairs == 1 rank x 6 pairs of the same 6 scenes"""
    if getattr(ctx, "simulated", False):
        pytest.skip("bench.py on the interpreter with two ranks is tests/test_sim_cpu.py")
    base = ["--config", "2", "--hits", "120000"]
    one, rec1 = _run(tmp_path, 1, base + ["--pairs-per-step", "6", "--distinct", "6"])
    two, rec2 = _run(tmp_path, 2, base + ["--pairs-per-step", "3", "--distinct", "3"])
    assert sorted(rec1) == sorted(rec2) == sorted(str(i) for i in range(6))
    for k in rec1:
        assert rec1[k] == rec2[k], "pair %s differs between one rank and two" % k
