#!/bin/bash
# Round 3, final GPU call: the whole GPU test suite, the default bench (cfg2) with the CPU legs and the parity check of all 64 scenes,
# and the other BASELINE configurations at full size (cfg4 with its CPU legs; cfg3 / cfg5 without -- their oracle runs take 65-200 s per
# pair and are held as committed fixtures, tests/golden/fullsize.json, which the GPU suite checks).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/r03_gputests_final.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r03_gputests_final.txt; tail -14 $O/r03_gputests_final.txt
timeout 1000 python bench.py --steps 3 --warmup 1 --scene-cache /tmp/scenes64 > $O/r03_bench_default.json 2> $O/r03_bench_default.err
echo "bench cfg2 rc=$?"; tail -c 4200 $O/r03_bench_default.json; cp $O/bench_detail_cfg2.json $O/r03_bench_default_detail.json
timeout 400 python bench.py --config 4 --steps 4 --warmup 1 > $O/r03_bench_cfg4.json 2> $O/r03_bench_cfg4.err
echo "bench cfg4 rc=$?"; tail -c 1500 $O/r03_bench_cfg4.json | cut -c1-1500
timeout 500 python bench.py --config 3 --steps 2 --warmup 1 --cpu-baseline 0 > $O/r03_bench_cfg3.json 2> $O/r03_bench_cfg3.err
echo "bench cfg3 rc=$?"; tail -c 1200 $O/r03_bench_cfg3.json | cut -c1-1200
timeout 600 python bench.py --config 5 --steps 1 --warmup 1 --pipeline 0 --cpu-baseline 0 > $O/r03_bench_cfg5.json 2> $O/r03_bench_cfg5.err
echo "bench cfg5 rc=$?"; tail -c 1200 $O/r03_bench_cfg5.json | cut -c1-1200
