"""TEST INFRASTRUCTURE: runs the repository's bench.py, unchanged, against the host SIMT interpreter (tests/hipsim) so that its host
logic -- front-end calibration, pipeline threads, batched front end, JSON line -- is exercised on a machine without a GPU.  torch's
CUDA entry points that bench.py touches are replaced by CPU stand-ins IN THIS PROCESS ONLY.  The numbers it prints mean nothing.

    python tests/hipsim/run_bench_sim.py --config 4 --hits 20000 --distinct 2 --pairs-per-step 4 --steps 1 --warmup 1 --cpu-baseline 0
"""
import importlib
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def patch_torch():
    """The interpreter's library behind api.Context and CPU stand-ins for the torch.cuda entry points the repository's scripts touch (this process only)."""
    import torch

    from hipsim import simctx

    api = importlib.import_module("gh-icp_amd.api")
    simctx.make_context(api).close()  # builds / loads the simulated library and replaces api.Context for this process

    class _Stream:
        cuda_stream = 0

    def _cpu_device(fn):
        def wrapped(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return fn(*a, **k)

        return wrapped

    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *_: None
    torch.cuda.synchronize = lambda *_: None
    torch.cuda.Stream = _Stream
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.zeros = _cpu_device(torch.zeros)
    torch.tensor = _cpu_device(torch.tensor)
    torch.empty = _cpu_device(torch.empty)


def main():
    patch_torch()
    if len(sys.argv) > 2 and sys.argv[1] == "--script":  # any other script of the repository, unchanged:  --script scripts/x.py [its arguments]
        script = os.path.join(ROOT, sys.argv[2])
        sys.argv = [script] + sys.argv[3:]
        runpy.run_path(script, run_name="__main__")
        return
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # one process per "GPU": RCCL becomes gloo, everything else of the N > 1 path is bench.py's own
        import torch.distributed as dist

        real_init = dist.init_process_group

        def init_gloo(backend=None, device_id=None, **kw):
            return real_init(backend="gloo", **kw)

        dist.init_process_group = init_gloo
    if "--detail-dir" not in sys.argv:  # the interpreter's side files stay out of gpurun_out/ (that directory holds MI355X records)
        sys.argv += ["--detail-dir", os.path.join(HERE, "_build", "bench_detail")]
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")


if __name__ == "__main__":
    main()
