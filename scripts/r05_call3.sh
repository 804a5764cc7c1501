#!/bin/bash
# Round 5, third GPU call (~9 GPU-minutes): PCA with the table lookups one cell ahead + flattened staging, cell table from the sorted keys,
# BSC sweeps four points at a time; counter passes that say what bounds the two big front-end kernels.
#   gpurun --timeout 900 -- 'bash scripts/r05_call3.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_cloud_cache.py tests/test_gpu_multirank.py tests/test_gpu_icp.py -m gpu -x -q --durations=4 > $O/r05_gputests_call3.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call3.txt; tail -8 $O/r05_gputests_call3.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o a -- $B1 > /dev/null 2> $O/r05_fe_call3.err
python $R/scripts/rocprof_summary.py /tmp/prof_h $O/r05_kernel_stats_fe_one_stream_call3.txt "front end on one stream (call 3): $B1" | head -16 | cut -c1-150
bash $R/scripts/r05_fe_pmc.sh
cd $R
timeout 400 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r05_bench_call3.json 2> $O/r05_bench_call3.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_call3.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), (d.get("batch_ms") or {}).get("front_end_ms_per_cloud_on_its_stream"))
PY
