#!/bin/bash
# Round 6, call 3: NMS rounds with wave-aggregated counters (call 2: one atomic per candidate on one address, 2.4 ms in the first round);
# CSR extents of flagged rows prefetched in the DFS and the flood.  Front end on one stream (new / base), five real matrices (new / base),
# default bench 3 steps (base / new).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_frontend.py tests/test_golden.py tests/test_gpu_loop.py -m gpu -q -x > $O/r06_gputests_call3.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call3.txt; tail -3 $O/r06_gputests_call3.txt
for v in new base; do
  L=""; [ $v = base ] && L="--lib $R/gh-icp_amd/libghicp_var_base.so"
  echo "== km_bench $v"; timeout 200 python scripts/km_bench.py --more --check $L 2>&1 | grep "^it" | tee -a $O/r06_km_variants_call3.txt
done
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
for v in new base; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o a -- $B1 > /dev/null 2> $O/r06_fe_call3_$v.err
  python $R/scripts/rocprof_summary.py /tmp/prof_$v $O/r06_kernel_stats_fe_one_stream_call3_$v.txt "front end on one stream (call 3, $v): $B1" | head -16 | cut -c1-150
done
cd $R
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call3_$v.json 2> $O/r06_bench_call3_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call3_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call3_$v.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
PY
done
