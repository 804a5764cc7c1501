#!/bin/bash
# Round 6, call 13: what a straggler is -- per pair of the last batch of each loop context: its longest solve, the iteration it belongs to, the CU the
# slot ran on (ghicp_ctx_loop_timeline, new bits).  Default bench, five times, until some batches show stragglers.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in a b c d e f; do
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call13_$v.json 2> $O/r06_bench_call13_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call13_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call13_$v.json").read().strip().splitlines()[-1])
p=d.get("pair_loop_stats") or {}
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, {k:p.get(k) for k in ("mean_solve_ms","longest_solve_ms","mean_launch_span_ms","idle_slot_fraction")})
t=json.load(open("gpurun_out/r06_bench_call13_${v}_detail.json"))["timeline"]
for b in t["last_batches"]:
    print("   span", b["span_s"], "pairs with a solve > 1 s:", b.get("pairs_whose_longest_solve_exceeds_1s"))
    for x in b["five_longest_solves"][:3]: print("      ", x)
PY
done
