#!/bin/bash
# Round 6, call 10: with the faster loop (lazy S), do the step's two knobs sit where they should?  --tail-fraction (when the next step's front ends
# start) and the margin on the confined class's share of the CUs.  Default bench, 3 steps each.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env, args
  env $2 timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 $3 > $O/r06_bench_call10_$1.json 2> $O/r06_bench_call10_$1.err
  cp $O/bench_detail_cfg2.json $O/r06_bench_call10_$1_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call10_$1.json").read().strip().splitlines()[-1])
p=d.get("pair_loop_stats") or {}
t=json.load(open("gpurun_out/r06_bench_call10_$1_detail.json"))["timeline"]
print("$1", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, {k:p.get(k) for k in ("mean_solve_ms","longest_solve_ms","mean_launch_span_ms","idle_slot_fraction")}, [b["span_s"] for b in t["last_batches"]])
print("   ", t["last_batches"][-1]["active_pairs_every_250ms"])
PY
}
run default "A=1" ""
run tail10 "A=1" "--tail-fraction 0.10"
run tail25 "A=1" "--tail-fraction 0.25"
run margin100 "GHICP_LOOP_CONFINE_MARGIN=1.0" ""
run margin085 "GHICP_LOOP_CONFINE_MARGIN=0.85" ""
run default2 "A=1" ""
