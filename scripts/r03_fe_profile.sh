#!/bin/bash
# Round 3, front-end kernel timing: GPU tests of the front end + an uncontended kernel trace (front end on one stream, batches of 32 clouds)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
mkdir -p $O
timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_cloud_cache.py -m gpu -x -q > $O/r03_gputests_9.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/r03_gputests_9.txt
(cd /tmp && export TMPDIR=/tmp && B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 512 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes" && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- $B1 > $O/r03_bench_rocprof_fe1_v6.json 2> $O/r03_rocprof_a6.err; python $R/scripts/rocprof_summary.py /tmp/prof_a $O/r03_kernel_stats_fe_one_stream_v6.txt "$B1" > /dev/null)
head -14 $O/r03_kernel_stats_fe_one_stream_v6.txt | cut -c1-150
tail -c 700 $O/r03_bench_rocprof_fe1_v6.json
