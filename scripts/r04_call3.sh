#!/bin/bash
# Round 4, third GPU call: queue order by cost hints (the 112-iteration pairs first), the PCA hit masks, two virtual ranks on one GPU
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_multirank.py tests/test_gpu_loop.py -m gpu -x -q --durations=5 > $O/r04_gputests_call3.txt 2>&1
echo "pytest rc=$?"; tail -12 $O/r04_gputests_call3.txt
SC="--scene-cache /tmp/scenes64"
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 $SC > $O/r04_bench_call3_hints.json 2> $O/r04_bench_call3_hints.err
echo "bench hints rc=$?"; tail -c 2600 $O/r04_bench_call3_hints.json; cp $O/bench_detail_cfg2.json $O/r04_bench_call3_hints_detail.json
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --queue-hints 0 $SC > $O/r04_bench_call3_nohints.json 2> $O/r04_bench_call3_nohints.err
echo "bench no hints rc=$?"; tail -c 1400 $O/r04_bench_call3_nohints.json; cp $O/bench_detail_cfg2.json $O/r04_bench_call3_nohints_detail.json
