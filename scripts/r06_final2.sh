#!/bin/bash
# Round 6, final GPU call (second edition: the tree with the library's own radix sort; scripts/r06_final.sh ran on the commit before it = call 20), most important first: the default bench (cfg2) with its CPU legs, the parity check of all 64
# scenes and the no-hints regions; the rocprofv3 kernel trace of the default command; FETCH_SIZE / WRITE_SIZE passes (counters only, one per
# run); cfg4 and the surveyed variants of cfg3 / cfg4 (--config 13 / 14) with their CPU legs; cfg5 with its CPU legs and parity check; the
# reference's call sequence through the drop-in headers (timing); and, LAST, the whole GPU suite in one command.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
SC="--scene-cache /tmp/scenes64"
timeout 900 python bench.py --steps 3 --warmup 1 $SC > $O/r06_bench_default.json 2> $O/r06_bench_default.err
echo "bench cfg2 rc=$?"; tail -c 3500 $O/r06_bench_default.json; cp $O/bench_detail_cfg2.json $O/r06_bench_default_detail.json
cd /tmp
D="python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 $SC"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o p -- $D > $O/r06_bench_default_under_rocprof.json 2> $O/r06_rocprof_kt.err
python $R/scripts/rocprof_summary.py /tmp/prof_kt $O/r06_kernel_stats_bench_default.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 (final tree)" | head -14 | cut -c1-150
python $R/scripts/rocprof_timeline.py /tmp/prof_kt $O/r06_timeline_bench_default.txt > /dev/null 2>&1
P="python $R/bench.py --steps 1 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 $SC"
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2; do   # (call 20: the FETCH_SIZE pass caught a SIGTERM two seconds after start and sat in its finalisation until the timeout)
    rm -rf /tmp/prof_$c
    timeout 330 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o p -- $P > $O/r06_pmc_${c}_bench.json 2> $O/r06_pmc_${c}.err
    rc=$?; echo "pmc $c attempt $attempt rc=$rc"
    [ $rc = 0 ] && break
  done
  python $R/scripts/rocprof_summary.py /tmp/prof_$c $O/r06_pmc_${c}.txt "pmc $c: python bench.py --steps 1 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 (final tree, default workload)" > /dev/null
  grep -h "k_pair_loop\|k_fb_pca_cells\|k_fb_bsc" $O/r06_pmc_${c}.txt | cut -c1-160
done
python $R/scripts/r06_pmc_traffic.py $O/r06_pmc_FETCH_SIZE.txt $O/r06_pmc_WRITE_SIZE.txt 5376 34.7 $O/r06_pmc_traffic.json
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 $SC"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fe -o a -- $B1 > /dev/null 2> $O/r06_fe_final.err
python $R/scripts/rocprof_summary.py /tmp/prof_fe $O/r06_kernel_stats_fe_one_stream_final.txt "front end on one stream (final tree): $B1" | head -16 | cut -c1-150
cd $R
timeout 200 python scripts/sort_bench.py 2>&1 | tail -4 | tee $O/r06_sort_bench_final.txt
timeout 400 python bench.py --config 4 --steps 4 --warmup 1 > $O/r06_bench_cfg4.json 2> $O/r06_bench_cfg4.err
echo "bench cfg4 rc=$?"; tail -c 1200 $O/r06_bench_cfg4.json
timeout 400 python bench.py --config 14 --steps 4 --warmup 1 > $O/r06_bench_cfg14.json 2> $O/r06_bench_cfg14.err
echo "bench cfg14 (cfg4 as surveyed) rc=$?"; tail -c 1200 $O/r06_bench_cfg14.json
timeout 700 python bench.py --config 13 --steps 2 --warmup 1 --cpu-procs 16 > $O/r06_bench_cfg13.json 2> $O/r06_bench_cfg13.err
echo "bench cfg13 (cfg3 as surveyed) rc=$?"; tail -c 1500 $O/r06_bench_cfg13.json
timeout 900 python bench.py --config 3 --steps 2 --warmup 1 --cpu-procs 16 > $O/r06_bench_cfg3.json 2> $O/r06_bench_cfg3.err
echo "bench cfg3 rc=$?"; tail -c 1500 $O/r06_bench_cfg3.json
timeout 1500 python bench.py --config 5 --steps 1 --warmup 1 --pipeline 0 --cpu-procs 8 --scene-cache /tmp/scenes5 > $O/r06_bench_cfg5.json 2> $O/r06_bench_cfg5.err
echo "bench cfg5 rc=$?"; tail -c 1500 $O/r06_bench_cfg5.json
timeout 300 python scripts/dropin_time.py > $O/r06_dropin_time.out 2> $O/r06_dropin_time.err; echo "dropin rc=$?"; tail -c 600 $O/r06_dropin_time.out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/r06_gputests_final.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_final.txt; tail -14 $O/r06_gputests_final.txt
