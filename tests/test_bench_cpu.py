"""bench.py's host-side helpers (no GPU): the byte models behind the roofline fields and the size guard of the ONE JSON line."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_timed_kernel_has_a_byte_model():
    fe = ("voxel_sort", "fb_voxel", "fb_grid", "pca_cells", "fb_prune", "fb_rank", "nms_round", "fb_out", "bsc")
    for k in bench.KERNELS:
        if k in fe:
            v = bench.front_end_bytes_per_cloud(k, 1_000_000, 225_000, 22_000, 657, 4, 2)
        elif k == "pair_loop":
            v = bench.pair_loop_bytes(657, 540_000, 34.7, 330)
        else:
            v = bench.algorithmic_bytes(657, 225_000, 540_000, 4, k, 92.5)
        assert v == v and v > 0, k  # round-2 verdict: no stage may print "GBps": null
    assert math.isnan(bench.front_end_bytes_per_cloud("nonexistent", 1, 1, 1, 1, 4, 2))


def test_pair_bytes_follows_survey_8d():
    # SURVEY.md 8(d'): per cloud S0 + S1 + S2 + S3, S4 once, I x (S5 + S6), S7 = 24 B per raw source point (main:153)
    n, m, c, ks, kt, V, it = 1_000_000, 500_000, 50_000, 4000, 4000, 4, 30
    b = bench.pair_bytes(n, m, c, ks, kt, V, it, True, 1500)
    s5 = 24 * 8000 + 2 * 16e6 + 12 * 8000 + 16 * 16e6
    assert abs(b - (2 * ((16 * n + 16 * m) + 36 * m + (20 * m + 4 * 4000) + (16 * m + 56 * 2.5 * 4000 + 48 * 4000)) + (56 * (4 * 4000 + 4000) + 2 * 16e6)
                    + it * (s5 + 48 * 1500 + 48 * 4000) + 24 * n)) < 1.0
    assert bench.pair_bytes(n, m, c, ks, kt, V, it, True, 1500, n_src=2 * n) == b + 24 * n  # S7 counts the raw SOURCE
    assert bench.pair_bytes(n, m, c, ks, kt, V, it, False, 1500) < b  # NN: no 16 n^2 per iteration
    assert bench.pair_loop_bytes(657, 540_000, 10, 300) * 2 == bench.pair_loop_bytes(657, 540_000, 20, 300)


def test_compact_keeps_the_line_below_the_drivers_tail():
    base = {"metric": "registered_pairs_per_sec", "value": 1.0, "roofline": {"frac": 0.1}, "cpu_baseline": {"value": 1}, "config": {"workload": "w"}}
    line = bench.compact(dict(base))
    assert json.loads(line) == base and " " not in line.replace("registered_pairs_per_sec", "")
    fat = dict(base, batch_ms={"x": "y" * 4000}, km_launch_stats={"z": "y" * 4000}, rank_wall_s={"per_rank": [0.0] * 500})
    out = json.loads(bench.compact(fat, limit=5000))
    assert len(bench.compact(fat, limit=5000)) <= 5000 and out["metric"] == base["metric"] and "roofline" in out and "cpu_baseline" in out


def test_config_table_matches_baseline_json():
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    # configs[0] is the CPU-runnable cfg1 (tests), 2..5 are bench lines; 13 / 14 = cfg3 / cfg4 with the geometry SURVEY.md §8d wrote
    assert len(cfgs) == 5 and sorted(bench.CONFIGS) == [2, 3, 4, 5, 13, 14]
    assert bench.CONFIGS[13]["base"] == 3 and bench.CONFIGS[14]["base"] == 4 and (bench.CONFIGS[14]["voxel"], bench.CONFIGS[14]["r"], bench.CONFIGS[14]["R"]) == (0.025, 0.10, 0.30)
    assert bench.CONFIGS[2]["feature"] == "BSC" and bench.CONFIGS[2]["corr"] == "KM" and bench.CONFIGS[2]["hits"] == 1_000_000 and bench.CONFIGS[2]["voxel"] == 0.1
    assert bench.CONFIGS[3]["feature"] == "FPFH" and bench.CONFIGS[3]["corr"] == "NNR" and bench.CONFIGS[3]["hits"] == 5_000_000
    assert bench.CONFIGS[4]["B"] == 64 and bench.CONFIGS[4]["scaling"] == "strong"
    assert bench.CONFIGS[5]["dof"] == 4 and bench.CONFIGS[5]["iou"] == 0.3 and bench.CONFIGS[5]["hits"] == 10_000_000
