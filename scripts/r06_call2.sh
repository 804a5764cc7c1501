#!/bin/bash
# Round 6, call 2: round-based whole-chip NMS + own scan / select / unique (prims.hip) + the C ABI pair queue.
# (1) the whole GPU suite; (2) the front end on ONE stream under rocprofv3, new library and the library of the commit before (base);
# (3) the default bench, 3 steps, base then new on the same box.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $O/r06_gputests_call2.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call2.txt; tail -12 $O/r06_gputests_call2.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
for v in new base; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o a -- $B1 > /dev/null 2> $O/r06_fe_call2_$v.err
  python $R/scripts/rocprof_summary.py /tmp/prof_$v $O/r06_kernel_stats_fe_one_stream_call2_$v.txt "front end on one stream (call 2, $v): $B1" | head -24 | cut -c1-150
done
cd $R
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call2_$v.json 2> $O/r06_bench_call2_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call2_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call2_$v.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
print({k:d["roofline"].get(k) for k in ("frac","frac_per_launch","avg_dispatch_ms","dispatches")})
PY
done
