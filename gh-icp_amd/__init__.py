"""gh-icp_amd: MI355X-native GH-ICP registration hot path (HIP kernels behind a C ABI).

The directory name carries a hyphen, so import it with
    importlib.import_module("gh-icp_amd")
(see `__graft_entry__.py`, `bench.py`, `tests/conftest.py`).
"""
