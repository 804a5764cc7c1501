#!/bin/bash
# Round 6, call 1: (1) the WHOLE GPU suite, no -x, on the tree with the de-randomised drop-in test; (2) cfg3 and cfg5 bench lines with their CPU legs
# on this box (round-5 verdict, next-round items 1 and 2).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=10 > $O/r06_gputests_call1.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call1.txt; tail -25 $O/r06_gputests_call1.txt
timeout 900 python bench.py --config 3 --steps 2 --warmup 1 --cpu-procs 16 > $O/r06_bench_cfg3.json 2> $O/r06_bench_cfg3.err
echo "bench cfg3 rc=$?"; tail -c 3000 $O/r06_bench_cfg3.json | cut -c1-3000; tail -5 $O/r06_bench_cfg3.err
timeout 1500 python bench.py --config 5 --steps 1 --warmup 1 --pipeline 0 --cpu-procs 8 --scene-cache /tmp/scenes5 > $O/r06_bench_cfg5.json 2> $O/r06_bench_cfg5.err
echo "bench cfg5 rc=$?"; tail -c 3000 $O/r06_bench_cfg5.json | cut -c1-3000; tail -5 $O/r06_bench_cfg5.err
