"""Fuzzes oracle/km_model.inc (the CPU model of the GPU Kuhn-Munkres state machine) against the reference traversal.\n    python scripts/km_model_fuzz.py SEED COUNT   -- 80 000 matrices (4 seeds x 20 000) ran clean in round 1."""
import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle as O

def gen(rng, n, kind):
    pen = rng.choice([5.0, 8.0, 20.0, 52.07])
    cd = rng.uniform(0, 3 * pen, (n, n))
    if kind == 0:      # GH-ICP-like: few candidates per row, many rows with none
        keep = rng.random((n, n)) < rng.choice([0.01, 0.05, 0.2])
        rows_none = rng.random(n) < rng.choice([0.0, 0.3, 0.7])
        keep[rows_none] = False
    elif kind == 1:    # dense-ish rows (> 64 explicit entries)
        keep = rng.random((n, n)) < 0.7
    elif kind == 2:    # quantised costs: many exact ties
        cd = np.round(cd * rng.choice([1, 2, 4])) / rng.choice([1, 2, 4])
        keep = rng.random((n, n)) < 0.3
    else:              # columns that nobody wants + duplicated rows
        keep = rng.random((n, n)) < 0.15
        keep[:, rng.random(n) < 0.4] = False
        dup = rng.integers(0, n, n // 3)
        cd[dup] = cd[(dup + 1) % n]; keep[dup] = keep[(dup + 1) % n]
    w = np.where(keep & (cd < pen), -cd, -pen)
    return w

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
bad = 0; t0 = time.time(); marched = 0; steps = 0
for t in range(N):
    n = int(rng.choice([3, 5, 8, 17, 40, 65, 100, 130, 200]))
    w = gen(rng, n, t % 4)
    ref, _ = O.km(w)
    for march, sweep, flood in ((True, False, False), (False, False, False), (True, True, False), (False, True, False), (True, True, True), (False, True, True)):
        m, s, mr, fp = O.km_model(w, march=march, sweep_first=sweep, flood_dead=flood)
        if not (m == ref).all():
            bad += 1
            np.save('/tmp/km_counterexample_%d.npy' % t, w)
            print('MISMATCH t', t, 'n', n, 'kind', t % 4, 'march', march, 'sweep_first', sweep, 'flood_dead', flood, flush=True)
        if march and not sweep: marched += mr; steps += s
print('matrices', N, 'mismatches', bad, 'marched share %.3f' % (marched / max(1, steps)), 'seconds %.1f' % (time.time() - t0))
