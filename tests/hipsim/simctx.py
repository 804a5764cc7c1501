"""TEST INFRASTRUCTURE: an api.Context on the host SIMT interpreter (tests/hipsim).  "Device" memory is host memory, so the tensors the
wrappers allocate live on torch's CPU device.  Only tests construct this (GHICP_SIM=1 or the sim_ctx fixture); the package cannot."""
import ctypes as C

from . import build as _build


def load_sim_library(api):
    lib = C.CDLL(_build.build())
    lib.ghicp_last_error.restype = C.c_char_p
    lib.ghicp_version.restype = C.c_char_p
    missing = [s for s in api.EXPORTS if not hasattr(lib, s)]
    assert not missing, missing
    return lib


def make_context(api):
    """Replaces api._lib and api.Context for this process (conftest does this only under GHICP_SIM=1, i.e. in a process that was
    started to run the GPU tests on the simulator)."""
    import torch

    lib = load_sim_library(api)
    api._lib = lib  # module-level helpers (default_params, icp_params, ...) go through api.load()

    class SimContext(api.Context):
        simulated = True

        def __init__(self, device=0, stream=None):
            self.torch = torch
            self.lib = lib
            self.device = 0
            h = C.c_void_p()
            rc = lib.ghicp_ctx_create(0, C.byref(h))
            if rc != 0:
                raise api.GhicpError("ghicp_ctx_create (hipsim) failed with code %d" % rc)
            self.h = h
            self.dev = torch.device("cpu")

        def set_stream(self, stream):
            pass

    api.Context = SimContext  # tests that build further contexts (worker streams) get simulated ones too
    return SimContext(0)


def counters(lib):
    out = (C.c_longlong * 8)()
    lib.hipsim_counters(out)
    keys = ("launches", "workgroups", "divergent_collectives", "memcpys", "memsets", "host_syncs", "library_calls", "mallocs")
    return dict(zip(keys, (int(v) for v in out)))
