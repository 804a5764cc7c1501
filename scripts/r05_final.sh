#!/bin/bash
# Round 5, final GPU call (~28 GPU-minutes) on the final tree, most important first: the whole GPU test suite; the default bench (cfg2) with the CPU
# legs, the parity check of all 64 scenes and the no-hints comparison regions; a rocprofv3 kernel trace of the default command; FETCH_SIZE /
# WRITE_SIZE passes of the default command (counters only, one per run); cfg4 with its CPU legs; cfg3 and cfg5 with their CPU legs ON THIS BOX (round-4 verdict: they had been timed on the build container).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
SC="--scene-cache /tmp/scenes64"
timeout 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/r05_gputests_final.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_final.txt; tail -14 $O/r05_gputests_final.txt
timeout 800 python bench.py --steps 3 --warmup 1 $SC > $O/r05_bench_default.json 2> $O/r05_bench_default.err
echo "bench cfg2 rc=$?"; tail -c 4500 $O/r05_bench_default.json; cp $O/bench_detail_cfg2.json $O/r05_bench_default_detail.json
cd /tmp
D="python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 $SC"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o p -- $D > $O/r05_bench_default_under_rocprof.json 2> $O/r05_rocprof_kt.err
python $R/scripts/rocprof_summary.py /tmp/prof_kt $O/r05_kernel_stats_bench_default.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-baseline 0 (final tree)" | head -12
python $R/scripts/rocprof_timeline.py /tmp/prof_kt $O/r05_timeline_bench_default.txt
P="python $R/bench.py --steps 1 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 $SC"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o p -- $P > $O/r05_pmc_${c}_bench.json 2> $O/r05_pmc_${c}.err
  python $R/scripts/rocprof_summary.py /tmp/prof_$c $O/r05_pmc_${c}.txt "pmc $c: python bench.py --steps 1 --warmup 1 --cpu-baseline 0 (final tree, default workload)" > /dev/null
  grep -h "k_pair_loop\|k_fb_pca_cells\|k_fb_bsc" $O/r05_pmc_${c}.txt | cut -c1-200
done
cd $R
timeout 400 python bench.py --config 4 --steps 4 --warmup 1 > $O/r05_bench_cfg4.json 2> $O/r05_bench_cfg4.err
echo "bench cfg4 rc=$?"; tail -c 1500 $O/r05_bench_cfg4.json | cut -c1-1500
cd $R
timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --cpu-procs 16 > $O/r05_bench_cfg3.json 2> $O/r05_bench_cfg3.err
echo "bench cfg3 rc=$?"; tail -c 2500 $O/r05_bench_cfg3.json | cut -c1-2500
timeout 900 python bench.py --config 5 --steps 1 --warmup 1 --pipeline 0 --cpu-procs 8 --scene-cache /tmp/scenes5 > $O/r05_bench_cfg5.json 2> $O/r05_bench_cfg5.err
echo "bench cfg5 rc=$?"; tail -c 2500 $O/r05_bench_cfg5.json | cut -c1-2500
