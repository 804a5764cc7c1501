#!/bin/bash
# Round 6, call 21: the library's own stable radix sort (prims.hip: k_rs_hist / k_rs_chunks / k_rs_bases / k_rs_scatter) in place of rocPRIM's Onesweep
# for the voxel keys, the grids' cell keys and the single-cloud NMS order.  Tests, front end on one stream (kernel trace, base / new), default bench base / new.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_golden.py tests/test_gpu_cloud_cache.py tests/test_gpu_zz_batch_fullsize.py tests/test_gpu_icp.py -m gpu -q -x > $O/r06_gputests_call21.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call21.txt; tail -3 $O/r06_gputests_call21.txt
timeout 200 python scripts/sort_bench.py 2>&1 | tail -6 | tee $O/r06_sort_bench_call21.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o a -- $B1 > /dev/null 2> $O/r06_fe_call21_$v.err
  python $R/scripts/rocprof_summary.py /tmp/prof_$v $O/r06_kernel_stats_fe_one_stream_call21_$v.txt "front end on one stream (call 21, $v): $B1" | head -24 | cut -c1-150
done
unset GHICP_LIB
cd $R
for v in base new base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  t=$RANDOM
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call21_${v}_$t.json 2> $O/r06_bench_call21_${v}_$t.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call21_${v}_${t}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call21_${v}_$t.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
t=json.load(open("gpurun_out/r06_bench_call21_${v}_${t}_detail.json"))
print(t["front_end_calibration"])
PY
done
