#!/bin/bash
# front end on one stream (uncontended kernel times) + the front-end GPU tests, for a front-end kernel change after the final call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o a -- $B1 > /dev/null 2> $O/r04_fe_check.err
python $R/scripts/rocprof_summary.py /tmp/prof_h $O/r04_fe_check.txt "front end on one stream: $B1" > /dev/null
grep -h "k_fb_bsc\|k_fb_pca_cells" $O/r04_fe_check.txt | cut -c1-150
cd $R
timeout 300 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not cfg5 and not cfg3" > $O/r04_gputests_fe_check.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r04_gputests_fe_check.txt; tail -3 $O/r04_gputests_fe_check.txt
