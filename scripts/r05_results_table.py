"""Prints the round-5 results from the bench lines under profiles/ (development aid)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    f = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(f):
        return None
    for l in reversed(open(f).read().strip().splitlines()):
        if l.startswith("{"):
            return json.loads(l)
    return None


for cfg, name in ((2, "r05_bench_default.json"), (3, "r05_bench_cfg3.json"), (4, "r05_bench_cfg4.json"), (5, "r05_bench_cfg5.json")):
    d = line(name)
    if not d:
        print("cfg%d: no line" % cfg)
        continue
    cpu = d.get("cpu_baseline") or {}
    chk = d.get("parity_check") or {}
    ro = d.get("registered_ok") or {}
    print("cfg%d: value %.2f / all %.2f pairs/s, %.1f ms/step, steps %d; no_hints %s; single pair %.3f s, %.2f ms/it; iterations mean %s %s; K mean %s, M mean %s"
          % (cfg, d["value"], d["value_all_pairs"], d["ms_per_step"], d["steps"], d.get("value_no_hints"), d["single_pair_latency_s"], d["ms_per_iteration"],
             d["config"].get("iterations_mean"), d["config"].get("iterations_min_max"), d["config"].get("k_mean"), d["config"].get("m_mean")))
    print("   verdict ok %s of %s per step, gt ok %s, gt failed scenes %s" % (ro.get("reference_verdict_ok"), ro.get("pairs_per_step_rank0"), ro.get("gt_ok"), ro.get("distinct_scenes_gt_failed")))
    print("   cpu 1thr %s pairs/s (%s), all cores %s; speedups %s / %s" % (cpu.get("value"), cpu.get("sample"), cpu.get("all_cores"), d.get("speedup_vs_cpu_1thread"), d.get("speedup_vs_cpu_all_cores")))
    print("   parity %s" % chk)
    r = d["roofline"]
    print("   roofline: kernel %s frac %s per-launch %s whole pair %s traffic_frac %s; pair_loop_stats %s" % (r["kernel"], r["frac"], r.get("frac_per_launch"), r["whole_pair_frac"], r.get("traffic_frac"), d.get("pair_loop_stats")))
    print("   batch_ms %s" % d.get("batch_ms"))
