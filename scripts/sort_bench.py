"""Times the library's own radix sort (prims.hip, ghicp_sort_pairs) at the front end's sizes on the GPU box: the packed voxel sort of a 32-cloud
batch (32 M u64 keys, 39 bits from bit 25, keys only) and a grid's cell sort (8 M u32 keys of 28 bits + u32 values).  Development aid."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
api = importlib.import_module("gh-icp_amd.api")
ctx = api.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
for name, n, kb, b, e, vals in (("voxel 32M u64 keys [25,64)", 32_000_000, 8, 25, 64, False), ("grid 8M u32 pairs [0,28)", 8_000_000, 4, 0, 28, True),
                                ("grid 8M u32 pairs [0,24)", 8_000_000, 4, 0, 24, True), ("cloud 1M u64 pairs [0,34)", 1_000_000, 8, 0, 34, True)):
    if kb == 8:
        k = torch.randint(0, 1 << 62, (n,), generator=g, device="cuda", dtype=torch.int64)
    else:
        k = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    v = torch.arange(n, device="cuda", dtype=torch.int32) if vals else None
    torch.cuda.synchronize()
    for _ in range(2):
        out = ctx.sort_pairs(k, v, b, e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        out = ctx.sort_pairs(k, v, b, e)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    passes = (e - b + 7) // 8
    bytes_ = n * passes * (kb * 3 + (8 if vals else 0))
    print("%-30s %8.3f ms  %d passes  %.2f TB/s of hist + scatter traffic" % (name, ms, passes, bytes_ / ms / 1e9), flush=True)
ctx.close()
