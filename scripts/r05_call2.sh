#!/bin/bash
# Round 5, second GPU call (~14 GPU-minutes): the tree after call 1 -- fused S rounds + packed voxel sort merged, PCA test / sum split + XCD-aware
# cell deal, BSC cell centres through LDS, staged-input cache, pinned job tables, cfg3 / cfg4 generators, RCCL one-rank group.
#   gpurun --timeout 1100 -- 'bash scripts/r05_call2.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
# (0) the tests that cover what changed: front end, batch, loop / solver, configs (cfg4 generator, staged cache), drop-in, RCCL group
timeout 700 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_loop.py tests/test_gpu_configs.py tests/test_gpu_dropin.py tests/test_gpu_multirank.py tests/test_gpu_cloud_cache.py -m gpu -x -q --durations=6 > $O/r05_gputests_call2.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call2.txt; tail -12 $O/r05_gputests_call2.txt
# (1) front end alone on one stream: uncontended kernel times (PCA, BSC, sorts)
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o a -- $B1 > /dev/null 2> $O/r05_fe_call2.err
python $R/scripts/rocprof_summary.py /tmp/prof_h $O/r05_kernel_stats_fe_one_stream_call2.txt "front end on one stream (call 2): $B1" | head -16 | cut -c1-150
cd $R
# (2) the bench line, 2 steps (with the no-hints comparison regions)
timeout 400 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes64 > $O/r05_bench_call2.json 2> $O/r05_bench_call2.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_call2.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step","value_no_hints","no_hints")}, d.get("pair_loop_stats"), d["roofline"].get("frac"), d["roofline"].get("frac_per_launch"))
PY
# (3) three slots per CU with the front ends issued early: does a quarter of every CU left free let the front end run beside the loop?
GHICP_LOOP_MIN_LDS=46080 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --tail-fraction 1.0 --scene-cache /tmp/scenes64 > $O/r05_bench_768_early.json 2> $O/r05_bench_768_early.err
echo "bench 768 slots, early front ends rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_768_early.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), (d.get("batch_ms") or {}).get("front_end_ms_per_cloud_on_its_stream"))
PY
# (4) the reference's own use case through the drop-in headers, 1 M points
timeout 200 python scripts/dropin_time.py 0 > $O/r05_dropin_time.log 2>&1; tail -c 1500 $O/r05_dropin_time.log
