#!/bin/bash
# Round 5, eighth GPU call (~4 GPU-minutes): class rule with LDS margin (four slots per CU must fit with room), class share margin 20 %, NMS back to round 4's.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_loop.py -m gpu -x -q > $O/r05_gputests_call8.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call8.txt; tail -3 $O/r05_gputests_call8.txt
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r05_bench_call8.json 2> $O/r05_bench_call8.err
echo "bench rc=$?"; cp $O/bench_detail_cfg2.json $O/r05_bench_call8_detail.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_call8.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
t=json.load(open("gpurun_out/r05_bench_call8_detail.json"))["timeline"]
print(t["loop_calls_s"]); print([ (b["span_s"], b["active_pairs_every_250ms"]) for b in t["last_batches"]])
PY
