#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun; outputs under gpurun_out/, summaries copied to profiles/ afterwards).
#   1. rocprofv3 kernel trace of a short bench with the front end on ONE stream (uncontended per-kernel durations of a batch of 32 clouds)
#   2. rocprofv3 kernel trace of the default schedule (the persistent pair loop's dispatches beside bench.py's own HIP-event time)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, counters only) of a small batch: front-end kernels + k_pair_loop
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
SC="--scene-cache /tmp/scenes"
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 512 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 $SC"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- $B1 > $O/r03_bench_rocprof_fe1.json 2> $O/r03_rocprof_a.err
python $R/scripts/rocprof_summary.py /tmp/prof_a $O/r03_kernel_stats_fe_one_stream.txt "$B1" > /dev/null
B2="python $R/bench.py --steps 2 --warmup 1 --distinct 16 --pairs-per-step 2048 --cpu-baseline 0 --pipeline 0 $SC"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- $B2 > $O/r03_bench_rocprof_default.json 2> $O/r03_rocprof_b.err
python $R/scripts/rocprof_summary.py /tmp/prof_b $O/r03_kernel_stats_bench.txt "$B2" > /dev/null
B3="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 $SC"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o p -- $B3 > $O/r03_pmc_${c}.json 2> $O/r03_pmc_${c}.err
  python $R/scripts/rocprof_summary.py /tmp/prof_$c $O/r03_pmc_${c}.txt "pmc $c: $B3" > /dev/null
done
head -40 $O/r03_kernel_stats_fe_one_stream.txt | cut -c1-160
grep -h "k_pair_loop" $O/r03_kernel_stats_bench.txt $O/r03_pmc_FETCH_SIZE.txt $O/r03_pmc_WRITE_SIZE.txt | cut -c1-200
tail -c 600 $O/r03_bench_rocprof_default.json
