#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun; outputs under gpurun_out/, summaries copied to profiles/ afterwards).
#   1. rocprofv3 kernel trace of a short default-schedule bench (8 distinct scenes, 448 pairs/step)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, counters only) of the three real cfg2 KM solves
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
BENCH="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 448 --cpu-baseline 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- $BENCH > $O/r02_bench_under_rocprof.json 2> $O/r02_rocprof_k.err
python $R/scripts/rocprof_summary.py /tmp/prof_k $O/r02_kernel_stats_bench.txt "$BENCH" > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o p -- python $R/scripts/km_bench.py > $O/r02_pmc_${c}.log 2>&1
  python $R/scripts/rocprof_summary.py /tmp/prof_$c $O/r02_pmc_${c}_km_solves.txt "pmc $c: scripts/km_bench.py (3 real cfg2 matrices, 2 solves each, n = 840)" > /dev/null
done
grep -h "k_km4" $O/r02_kernel_stats_bench.txt $O/r02_pmc_FETCH_SIZE_km_solves.txt $O/r02_pmc_WRITE_SIZE_km_solves.txt
