#!/bin/bash
# Round 4, fourth GPU call: hinted queue order with the class LDS fix, BSC with the contract expf + exact depth sums (strings must equal the oracle's)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_loop.py tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=5 -k "not cfg5 and not cfg3" > $O/r04_gputests_call4.txt 2>&1
echo "pytest rc=$?"; tail -12 $O/r04_gputests_call4.txt
SC="--scene-cache /tmp/scenes64"
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 $SC > $O/r04_bench_call4_hints.json 2> $O/r04_bench_call4_hints.err
echo "bench hints rc=$?"; tail -c 2600 $O/r04_bench_call4_hints.json; cp $O/bench_detail_cfg2.json $O/r04_bench_call4_hints_detail.json; tail -5 $O/r04_bench_call4_hints.err
