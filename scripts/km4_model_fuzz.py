"""Fuzzes oracle/km4_model.inc (the rule-level model of the flood-first Kuhn-Munkres kernel k_km4) against the
reference traversal (orc::KM, itself pinned to the reference's compiled km.cpp by tests/test_oracle_cpu.py).
    python scripts/km4_model_fuzz.py SEED COUNT
Generators 0-3 are those of scripts/km_model_fuzz.py; 4 puts the costs on a coarse lattice +- a few ulp, so that label sums
that are equal in exact arithmetic differ in the last bits.  (Costs within ulps of eps-tightness -- the case rule R4's hazard
check exists for -- cannot be fuzzed against the reference: km.cpp itself does not terminate on such matrices, its delta
becomes 0 when a visited column drops out.)"""
import sys, time, numpy as np
sys.path.insert(0, __file__.rsplit('/', 2)[0])
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
    elif kind == 3:    # columns that nobody wants + duplicated rows
        keep = rng.random((n, n)) < 0.15
        keep[:, rng.random(n) < 0.4] = False
        dup = rng.integers(0, n, n // 3)
        cd[dup] = cd[(dup + 1) % n]; keep[dup] = keep[(dup + 1) % n]
    else:              # costs on a lattice of spacing 0.25 +- a few ulp
        base = np.round(rng.uniform(0, pen, (n, n)) / 0.25) * 0.25
        cd = base + rng.integers(-3, 4, (n, n)) * np.spacing(base)
        keep = rng.random((n, n)) < rng.choice([0.1, 0.4])
    return np.where(keep & (cd < pen), -cd, -pen)


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    bad = hazards = 0
    t0 = time.time()
    for t in range(N):
        n = int(rng.choice([3, 5, 8, 17, 40, 65, 100, 130, 200]))
        w = gen(rng, n, t % 5)
        ref, _ = O.km(w)
        for cap, prune, kw in ((1, True, {}), (3, True, {}), (3, False, {}), (6, True, {}), (3, True, dict(hint=6, exact_rest=True)),
                               (3, True, dict(hint=6, exact_rest=True, seed=True)), (3, True, dict(hint=6, exact_rest=True, seed=True, lazy=True)), (1, True, dict(lazy=True)),
                               (3, True, dict(hint=6, lazy=True))):  # the kernel's rules R5, R3' (self-checked) and R5' (lazy S)
            m, _st = O.km4_model(w, cap=cap, prune=prune, **kw)
            if m is None:
                hazards += 1
                continue
            if not (m == ref).all():
                bad += 1
                np.save('/tmp/km4_counterexample_%d.npy' % t, w)
                print('MISMATCH t', t, 'n', n, 'kind', t % 5, 'cap', cap, 'prune', prune, kw, flush=True)
    print('matrices', N, 'mismatches', bad, 'hazard reports', hazards, 'seconds %.1f' % (time.time() - t0))


if __name__ == '__main__':
    main()
