#!/bin/bash
# Round 4, last GPU call: the BSC kernel changed after the final suite run (depth sums through a wrapping 64-bit word + carry instead of two
# integer atomics; ring tests behind a pre-filter): its GPU tests again, its uncontended time, and the default bench line on this tree
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py tests/test_gpu_cloud_cache.py -m gpu -x -q -k "not cfg5 and not cfg3" > $O/r04_gputests_after_final.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r04_gputests_after_final.txt; tail -4 $O/r04_gputests_after_final.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o a -- $B1 > /dev/null 2> $O/r04_bscvar_head2.err
python $R/scripts/rocprof_summary.py /tmp/prof_h $O/r04_bscvar_head2.txt "BSC at HEAD (pre-filtered ring): $B1" > /dev/null
grep -h "k_fb_bsc\|k_fb_pca_cells" $O/r04_bscvar_head2.txt | cut -c1-150
cd $R
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes64 > $O/r04_bench_head.json 2> $O/r04_bench_head.err
echo "bench rc=$?"; tail -c 2200 $O/r04_bench_head.json | cut -c1-2200
