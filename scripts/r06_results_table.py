"""The rows of BASELINE.md §3 / DESIGN.md §7 from the bench lines of the round's final call (profiles/r06_bench_*.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    rows = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
    return json.loads(rows[-1]) if rows else None


def main():
    print("| cfg | GPUs | pairs/s registered / all | step | single pair, ms/iter | B_alg per pair (GB) | dominant kernel: % of HBM roofline (span / per dispatch) | CPU 1-thr pairs/s | CPU N-process pairs/s | speed-up 1-thr / N-proc (registered) | parity vs oracle |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for cfg, f in ((2, "r06_bench_default.json"), (3, "r06_bench_cfg3.json"), (13, "r06_bench_cfg13.json"), (4, "r06_bench_cfg4.json"), (14, "r06_bench_cfg14.json"), (5, "r06_bench_cfg5.json")):
        d = line(f)
        if d is None:
            print("| %d | -- missing %s |" % (cfg, f))
            continue
        r, c, p, cf = d["roofline"], d.get("cpu_baseline") or {}, d.get("parity_check") or {}, d["config"]
        ac = c.get("all_cores") or {}
        reg = d.get("registered_ok") or {}
        print("| %s | 1 | **%.1f / %.1f**%s; verdict accepts %s of %s pairs per step, %s within 0.05 rot / 0.5 m of ground truth | %.0f ms, %d pairs | %.2f s, %.1f ms/iter (I %s) | %.3f | `%s` %.2f %% / %s | %s | %s (%s proc) | %s x / %s x | %s |" % (
            cf["workload"].split(":")[0], d["value"], d["value_all_pairs"], (" (no queue-order prior: %.1f / %.1f)" % (d["value_no_hints"], d["value_all_pairs_no_hints"])) if d.get("value_no_hints") else "",
            reg.get("reference_verdict_ok"), reg.get("pairs_per_step_rank0"), reg.get("gt_ok"), d["ms_per_step"], cf["pairs_per_step"], d.get("single_pair_latency_s", float("nan")),
            d.get("ms_per_iteration", float("nan")), cf.get("iterations_min_max"), r["alg_bytes_per_pair"] / 1e9, r["kernel"], 100 * r["frac"],
            ("%.2f %%" % (100 * r["frac_per_launch"])) if r.get("frac_per_launch") is not None else "--", c.get("value"), ac.get("value"), ac.get("cores"),
            d.get("speedup_vs_cpu_1thread"), d.get("speedup_vs_cpu_all_cores"),
            ("%s of %s pairs: iterations / keypoints / verdict identical, rot %s, trans %s m" % (p.get("iterations_match"), p.get("pairs_checked"), p.get("max_rot_err_vs_oracle"), p.get("max_trans_err_vs_oracle_m"))) if p.get("pairs_checked") else (p.get("note") or "--")))


if __name__ == "__main__":
    main()
