#!/bin/bash
# Round 5, fifth GPU call (~5 GPU-minutes): one Kuhn-Munkres class for every graph that fits four per CU (one queue in cost order).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_loop.py tests/test_gpu_multirank.py -m gpu -x -q > $O/r05_gputests_call5.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call5.txt; tail -4 $O/r05_gputests_call5.txt
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes64 > $O/r05_bench_call5.json 2> $O/r05_bench_call5.err
echo "bench rc=$?"; cp $O/bench_detail_cfg2.json $O/r05_bench_call5_detail.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_call5.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step","value_no_hints","no_hints")}, d.get("pair_loop_stats"), d.get("batch_ms"))
t=json.load(open("gpurun_out/r05_bench_call5_detail.json"))["timeline"]
print(t["loop_calls_s"]); print([ (b["span_s"], b["active_pairs_every_250ms"]) for b in t["last_batches"]])
PY
