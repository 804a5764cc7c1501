#!/bin/bash
# Round 5, last GPU call (the 10 GPU-minutes that were left): the final call's bench lines died in a format string edited after the last dry run
# (bench.py:919, TypeError) -- tests, kernel trace and counter passes of that call stand (profiles/r05_final_log.txt); here the lines themselves.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --steps 3 --warmup 1 --no-hints-steps 1 --scene-cache /tmp/scenes64 > $O/r05_bench_default.json 2> $O/r05_bench_default.err
echo "bench cfg2 rc=$?"; tail -c 3000 $O/r05_bench_default.json; cp $O/bench_detail_cfg2.json $O/r05_bench_default_detail.json
timeout 80 python bench.py --config 4 --steps 4 --warmup 1 > $O/r05_bench_cfg4.json 2> $O/r05_bench_cfg4.err
echo "bench cfg4 rc=$?"; tail -c 1500 $O/r05_bench_cfg4.json | cut -c1-1500
timeout 100 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $O/r05_gputests_last.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_last.txt; tail -3 $O/r05_gputests_last.txt
