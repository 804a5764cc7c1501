#!/bin/bash
# Round 3, first GPU call: the GPU test suite (incl. the new full-size 4x4 fixtures, the 64 cfg4 pairs against the live oracle, the FPFH
# branch of the batched front end, the persistent pair loop) and a short default bench.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/r03_gputests_1.txt 2>&1
echo "pytest rc=$?" >> $O/r03_gputests_1.txt
tail -30 $O/r03_gputests_1.txt
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $O/r03_bench_quick1.json 2> $O/r03_bench_quick1.err
echo "bench rc=$?"
tail -c 3000 $O/r03_bench_quick1.json
tail -5 $O/r03_bench_quick1.err
