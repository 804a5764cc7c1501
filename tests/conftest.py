import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("gh-icp_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O  # test infrastructure (CPU checker)

    O.build()
    return O


@pytest.fixture(scope="session")
def api():
    return importlib.import_module("gh-icp_amd.api")


@pytest.fixture(scope="session")
def ctx(api):
    import torch

    if os.environ.get("GHICP_SIM") == "1":  # kernel development aid: the same tests on the host SIMT interpreter (tests/hipsim), never on the GPU box
        from hipsim import simctx

        c = simctx.make_context(api)
        yield c
        c.close()
        return
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = api.Context(0)
    yield c
    c.close()


def rot_err(A, B):
    return float(np.linalg.norm(A[:3, :3] @ B[:3, :3].T - np.eye(3)))


def trans_err(A, B):
    return float(np.linalg.norm(A[:3, 3] - B[:3, 3]))
