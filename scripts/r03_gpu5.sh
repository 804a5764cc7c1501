#!/bin/bash
# Round 3, GPU call 5: front-end kernels after the BSC sphere compaction and the PCA lane split (tests, uncontended kernel trace), then
# the default bench with the CPU legs / parity of all 64 scenes.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_cloud_cache.py -m gpu -x -q > $O/r03_gputests_5.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/r03_gputests_5.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg2 or cfg4 or properties" >> $O/r03_gputests_5.txt 2>&1
echo "pytest fullsize rc=$?"; tail -3 $O/r03_gputests_5.txt
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 512 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes" && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- $B1 > $O/r03_bench_rocprof_fe1_v2.json 2> $O/r03_rocprof_a2.err; python $R/scripts/rocprof_summary.py /tmp/prof_a $O/r03_kernel_stats_fe_one_stream_v2.txt "$B1" > /dev/null)
head -12 $O/r03_kernel_stats_fe_one_stream_v2.txt | cut -c1-150
timeout 900 python bench.py --steps 3 --warmup 1 --cpu-procs 64 --scene-cache /tmp/scenes64 > $O/r03_bench_v5.json 2> $O/r03_bench_v5.err
echo "bench rc=$?"
tail -c 3800 $O/r03_bench_v5.json
tail -3 $O/r03_bench_v5.err
