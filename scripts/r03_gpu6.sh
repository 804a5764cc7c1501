#!/bin/bash
# Round 3, GPU call 6: front-end kernels after the cell-centric BSC sweep and the resident PCA tile (tests, uncontended kernel trace, short bench)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
mkdir -p $O
timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_cloud_cache.py -m gpu -x -q > $O/r03_gputests_7.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/r03_gputests_7.txt
(cd /tmp && export TMPDIR=/tmp && B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 512 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes" && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- $B1 > $O/r03_bench_rocprof_fe1_v4.json 2> $O/r03_rocprof_a4.err; python $R/scripts/rocprof_summary.py /tmp/prof_a $O/r03_kernel_stats_fe_one_stream_v4.txt "$B1" > /dev/null)
head -14 $O/r03_kernel_stats_fe_one_stream_v4.txt | cut -c1-150
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes64 > $O/r03_bench_v7.json 2> $O/r03_bench_v7.err
echo "bench rc=$?"
python - <<EOF
import json
d=json.loads(open("$O/r03_bench_v7.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["batch_ms"], d["pair_loop_stats"], d["roofline"]["per_kernel_GBps"])
EOF
