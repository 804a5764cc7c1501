#!/bin/bash
# Round 6, call 25 (no source change since the final call): smoke(), three consecutive default benches (run-to-run spread, the longest solves), the bench's
# kernel trace with the front-end shape FIXED at 64 clouds x 2 streams for the library before the own radix sort and the final one (comparable rows), and,
# LAST again, the whole GPU suite in one command.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/r06_smoke_call25.txt
SC="--scene-cache /tmp/scenes64"
for i in 1 2 3; do
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 $SC > $O/r06_bench_call25_run$i.json 2> $O/r06_bench_call25_run$i.err
  echo "bench run $i rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call25_run${i}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call25_run$i.json").read().strip().splitlines()[-1])
t=json.load(open("gpurun_out/r06_bench_call25_run${i}_detail.json"))
print("run $i", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), t["front_end_calibration"]["chosen"])
PY
done
cd /tmp
D="python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 64 --fe-batch-streams 2 $SC"
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt_$v -o p -- $D > $O/r06_bench_call25_under_rocprof_$v.json 2> $O/r06_rocprof_call25_$v.err
  python $R/scripts/rocprof_summary.py /tmp/prof_kt_$v $O/r06_kernel_stats_bench_64x2_$v.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 64 --fe-batch-streams 2 (call 25, library: $v)" | head -12 | cut -c1-150
done
unset GHICP_LIB
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/r06_gputests_call25.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call25.txt; tail -5 $O/r06_gputests_call25.txt
