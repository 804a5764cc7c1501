#!/bin/bash
# Round 6, call 9: are the multi-second solves of a batch solves that took the literal fallback (ghicp_ctx_loop_hazards)?  Default bench, new library, twice.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in a b; do
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call9_$v.json 2> $O/r06_bench_call9_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call9_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call9_$v.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"))
t=json.load(open("gpurun_out/r06_bench_call9_${v}_detail.json"))["timeline"]
for b in t["last_batches"]: print(b["span_s"], [(x["begin_s"],x["end_s"],x["iterations"]) for x in b["ten_longest"][:3]])
PY
done
