"""Rule-level model of the flood-first Kuhn-Munkres kernel (oracle/km4_model.inc) against the restatement of the reference
traversal (oracle orc::KM, pinned to the reference's compiled km.cpp in test_oracle_cpu.py) -- CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import km4_model_fuzz as F  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("it", [0, 10, 30])
def test_km4_model_real_matrices(oracle, it):
    z = np.load(os.path.join(GOLD, "km_cfg2_it%d.npz" % it))
    n = int(z["n"])
    w = np.full((n, n), float(z["bg"]))
    w[z["rows"].astype(np.int64), z["cols"].astype(np.int64)] = z["vals"]
    ref, _ = oracle.km(w)
    for cap in (1, 3):
        m, st = oracle.km4_model(w, cap=cap)
        assert m is not None, "hazard on a real matrix"
        np.testing.assert_array_equal(m, ref)
        assert st["failed"] in (727, 3193, 784) and (cap != 3 or st["dfs_steps"] < 200_000)
    # the kernel's treatment of flagged rows (hints for 4..6 tight entries, the others tested exactly): same matching, and the
    # activation / round counts are the ones k_km4 itself reports on these matrices (GHICP_KM_STATS, profiles/r03_km4_second_half.txt)
    m, st = oracle.km4_model(w, cap=3, hint=6, exact_rest=True)
    np.testing.assert_array_equal(m, ref)
    assert (st["dfs_steps"], st["pull_rounds"]) == {0: (156555, 4324), 10: (14409, 4208), 30: (58440, 2928)}[it]
    base = oracle.km4_model(w, cap=3)[1]
    assert st["dfs_pops"] * 2 < base["dfs_pops"]  # what the rule is for: the pops out of flagged rows that lead nowhere
    # design groundwork (not in the kernel): the seeded flood of rule R3' -- same matching, every failed-phase flood certified,
    # the closure checked against a flood from the root inside the model, and what it is for: the rows a flood has to visit
    m2, s2 = oracle.km4_model(w, cap=3, hint=6, exact_rest=True, seed=True)
    np.testing.assert_array_equal(m2, ref)
    assert s2["unseeded"] == 0 and s2["seeded"] == s2["failed"] and s2["flood_rows"] * 1.9 < st["flood_rows"]
    assert all(s2[k] == st[k] for k in ("phases", "failed", "push_rows", "rebuild_rows", "pull_rounds", "dfs_steps"))


@pytest.mark.parametrize("name,counts", [("s22_it46", (26328, 8446, 6943)), ("s53_it0", (354958, 5176, 424))])
def test_km4_model_heaviest_and_largest_matrix(oracle, name, counts):
    """The heaviest (scene 22, iteration 46: 6943 failed phases) and the largest (scene 53, n = 1131) of the 2220 matrices the 64 bench
    scenes solve (scripts/km_hazard_survey.py): no hazard, the reference's matching, the counts the survey reported."""
    z = np.load(os.path.join(GOLD, "km_cfg2_%s.npz" % name))
    n = int(z["n"])
    w = np.full((n, n), float(z["bg"]))
    w[z["rows"].astype(np.int64), z["cols"].astype(np.int64)] = z["vals"]
    m, st = oracle.km4_model(w, cap=3, hint=6, exact_rest=True)
    assert m is not None
    np.testing.assert_array_equal(m, oracle.km(w)[0])
    assert (st["dfs_steps"], st["pull_rounds"], st["failed"]) == counts


def test_km4_model_fuzz(oracle):
    rng = np.random.default_rng(20260925)
    for t in range(400):
        n = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 65, 100, 130]))
        w = F.gen(rng, n, t % 5)
        ref, _ = oracle.km(w)
        for cap, prune in ((1, True), (3, True), (3, False)):
            m, _ = oracle.km4_model(w, cap=cap, prune=prune)
            assert m is not None
            np.testing.assert_array_equal(m, ref, err_msg="t=%d n=%d cap=%d prune=%s" % (t, n, cap, prune))
        for kw in (dict(hint=6, exact_rest=True), dict(hint=6), dict(exact_s=True), dict(cap=1, hint=2, exact_rest=True),
                   dict(hint=6, exact_rest=True, seed=True), dict(prune=False, seed=True)):  # seed: R3', self-checked (status 7 raises)
            m, _ = oracle.km4_model(w, **kw)  # returns status 6 (raises) if the best column of S ever has a label other than 0
            assert m is not None
            np.testing.assert_array_equal(m, ref, err_msg="t=%d n=%d %r" % (t, n, kw))


def test_km4_model_kat_and_degenerate(oracle):
    W = np.array([[-5, -2, -100], [-4, -2, -6], [-100, -1, -7]], np.float64)  # km.cpp:237-259
    assert oracle.km4_model(W)[0].tolist() == [0, 2, 1]
    for w in (np.full((7, 7), -3.0), np.zeros((1, 1)), -np.eye(6) * 2.0 - 1.0):
        np.testing.assert_array_equal(oracle.km4_model(w)[0], oracle.km(w)[0])


@pytest.mark.parametrize("name,lazy_none", [("it0", 177), ("it10", 402), ("it30", 349), ("s22_it46", 206), ("s53_it0", 262)])
def test_km4_model_lazy_s_real_matrices(oracle, name, lazy_none):
    """Rule R5' (round 6, the kernel's configuration since): the search of an augmenting phase starts WITHOUT the set S and S is computed
    when the search first steps back; the stack is then cut back to its deepest good frame.  Same matching as the reference on the five
    real matrices; 21-48 % of their augmenting phases need no S at all, the pull rounds drop accordingly, and the search pays ~1 % more
    activations for its excursions into rows that are not good."""
    z = np.load(os.path.join(GOLD, "km_cfg2_%s.npz" % name))
    n = int(z["n"])
    w = np.full((n, n), float(z["bg"]))
    w[z["rows"].astype(np.int64), z["cols"].astype(np.int64)] = z["vals"]
    ref = oracle.km(w)[0]
    m0, s0 = oracle.km4_model(w, cap=3, hint=6, exact_rest=True, seed=True)
    m1, s1 = oracle.km4_model(w, cap=3, hint=6, exact_rest=True, seed=True, lazy=True)
    assert m0 is not None and m1 is not None
    np.testing.assert_array_equal(m0, ref)
    np.testing.assert_array_equal(m1, ref)
    assert s1["lazy_none"] == lazy_none and 0.2 * (s1["phases"] - s1["failed"]) < lazy_none < 0.5 * (s1["phases"] - s1["failed"])
    assert s1["pull_rounds"] < s0["pull_rounds"] and s0["dfs_steps"] <= s1["dfs_steps"] < 1.1 * s0["dfs_steps"]
    assert all(s1[k] == s0[k] for k in ("phases", "failed", "flood_rows", "push_rows", "rebuild_rows", "dfs_pops"))


def test_km4_model_lazy_s_fuzz(oracle):
    sys_path = os.path.join(ROOT, "scripts")
    import sys

    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import km4_model_fuzz as F

    rng = np.random.default_rng(20260930)
    none = 0
    for t in range(300):
        n = int(rng.choice([3, 5, 8, 17, 40, 65, 100, 130]))
        w = F.gen(rng, n, t % 5)
        ref, _ = oracle.km(w)
        for kw in (dict(cap=3, hint=6, exact_rest=True, seed=True, lazy=True), dict(cap=1, lazy=True), dict(cap=3, hint=6, lazy=True)):
            m, st = oracle.km4_model(w, **kw)
            assert m is not None
            np.testing.assert_array_equal(m, ref)
            none += st["lazy_none"]
    assert none > 0
