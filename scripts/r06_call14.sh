#!/bin/bash
# Round 6, call 14: the stragglers stall together (call 13: three pairs of one batch, each ONE solve of 8.4 s at iteration 8, all on die 7, resumed when the
# batch drained) -- queue preemption under oversubscribed hardware queues?  bench.py asks for GPU_MAX_HW_QUEUES=16 and the loop adds masked streams
# (own queues).  Default bench with 8 and with 4 hardware queues, four times each: do the stragglers go away, and what does the front end lose?
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for q in 8 4 8 4 8 4 8 4; do
  v=q${q}_$RANDOM
  GPU_MAX_HW_QUEUES=$q timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call14_$v.json 2> $O/r06_bench_call14_$v.err
  cp $O/bench_detail_cfg2.json $O/r06_bench_call14_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call14_$v.json").read().strip().splitlines()[-1])
p=d.get("pair_loop_stats") or {}
t=json.load(open("gpurun_out/r06_bench_call14_${v}_detail.json"))
print("hw queues $q", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, {k:p.get(k) for k in ("mean_solve_ms","longest_solve_ms","mean_launch_span_ms","idle_slot_fraction")}, t["front_end_calibration"]["chosen"], t["front_end_calibration"]["batched_clouds_per_s"])
for b in t["timeline"]["last_batches"]:
    if b.get("pairs_whose_longest_solve_exceeds_1s"): print("   span", b["span_s"], "pairs with a solve > 1 s:", b.get("pairs_whose_longest_solve_exceeds_1s"), b["five_longest_solves"][:2])
PY
done
