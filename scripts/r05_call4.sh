#!/bin/bash
# Round 5, fourth GPU call (~6 GPU-minutes): PCA runs handed out dynamically (per-XCD counters), BSC iterator reverted.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_multirank.py -m gpu -x -q --durations=4 > $O/r05_gputests_call4.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call4.txt; tail -6 $O/r05_gputests_call4.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o a -- $B1 > /dev/null 2> $O/r05_fe_call4.err
python $R/scripts/rocprof_summary.py /tmp/prof_h $O/r05_kernel_stats_fe_one_stream_call4.txt "front end on one stream (call 4): $B1" | head -14 | cut -c1-150
G="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/prof_w -o p -- $B1 > /dev/null 2> $O/r05_fe_pmc_call4.err
python $R/scripts/rocprof_summary.py /tmp/prof_w $O/r05_fe_pmc_call4_waves.txt "pmc $G (call 4)" > /dev/null
grep -h "k_fb_pca_cells\|k_fb_bsc" $O/r05_fe_pmc_call4_waves.txt | cut -c1-200
cd $R
timeout 400 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r05_bench_call4.json 2> $O/r05_bench_call4.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_call4.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), (d.get("batch_ms") or {}).get("front_end_ms_per_cloud_on_its_stream"))
PY
