#!/bin/bash
# Round 3, GPU call 2: GPU test suite, the three real Kuhn-Munkres matrices alone (timing + stage counters), a short default bench.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/r03_gputests_2.txt 2>&1
echo "pytest rc=$?" >> $O/r03_gputests_2.txt
tail -30 $O/r03_gputests_2.txt
(timeout 120 python scripts/km_bench.py --check; GHICP_KM_STATS=1 timeout 120 python scripts/km_bench.py) > $O/r03_km_bench.txt 2>&1
cat $O/r03_km_bench.txt
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $O/r03_bench_quick2.json 2> $O/r03_bench_quick2.err
echo "bench rc=$?"
tail -c 3500 $O/r03_bench_quick2.json
tail -5 $O/r03_bench_quick2.err
