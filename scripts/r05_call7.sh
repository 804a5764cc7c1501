#!/bin/bash
# Round 5, seventh GPU call (~6 GPU-minutes): cost prior iterations x n (class shares), NMS with 1024-candidate chunks.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_loop.py -m gpu -x -q > $O/r05_gputests_call7.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call7.txt; tail -3 $O/r05_gputests_call7.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o a -- $B1 > /dev/null 2> $O/r05_fe_call7.err
python $R/scripts/rocprof_summary.py /tmp/prof_h $O/r05_kernel_stats_fe_one_stream_call7.txt "front end on one stream (call 7): $B1" | head -12 | cut -c1-150
cd $R
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r05_bench_call7.json 2> $O/r05_bench_call7.err
echo "bench rc=$?"; cp $O/bench_detail_cfg2.json $O/r05_bench_call7_detail.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_call7.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
t=json.load(open("gpurun_out/r05_bench_call7_detail.json"))["timeline"]
print(t["loop_calls_s"]); print([ (b["span_s"], b["active_pairs_every_250ms"]) for b in t["last_batches"]])
PY
