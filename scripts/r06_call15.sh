#!/bin/bash
# Round 6, call 15: the stragglers are simultaneous stalls of a few slots on die 7 (shader engines 2 / 3) that end when the batch drains (calls 13, 14),
# with 16, 8 and 4 hardware queues alike.  Are the CU-masked queues of the confinement involved?  Default bench with GHICP_LOOP_CONFINE=0 (no masked
# streams), five times, and with two hardware queues, three times.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
run() {
  v=$1_$RANDOM
  env $2 timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call15_$v.json 2> $O/r06_bench_call15_$v.err
  cp $O/bench_detail_cfg2.json $O/r06_bench_call15_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call15_$v.json").read().strip().splitlines()[-1])
p=d.get("pair_loop_stats") or {}
t=json.load(open("gpurun_out/r06_bench_call15_${v}_detail.json"))
print("$1", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, {k:p.get(k) for k in ("mean_solve_ms","longest_solve_ms","mean_launch_span_ms","idle_slot_fraction")})
for b in t["timeline"]["last_batches"]:
    if b.get("pairs_whose_longest_solve_exceeds_1s"): print("   span", b["span_s"], "pairs with a solve > 1 s:", b.get("pairs_whose_longest_solve_exceeds_1s"), b["five_longest_solves"][:2])
PY
}
for i in 1 2 3 4 5; do run noconfine "GHICP_LOOP_CONFINE=0"; done
for i in 1 2 3; do run q2 "GPU_MAX_HW_QUEUES=2"; done
