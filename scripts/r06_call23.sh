#!/bin/bash
# Round 6, call 23: (1) the radix sort without k_rs_bases up to 32 chunks (the scatter sums the chunk rows itself): tests, timing, cfg4 (launch bound: 64 pairs of
# 100 k points, a 14 ms step) base / new with the front-end shape FIXED (call 22's cfg4 lines differ in the shape the calibration picked);
# (2) with the look-back sort gone, does an earlier start of the next step's front ends (--tail-fraction) pay?  (call 10: 0.10 / 0.25 within noise or worse)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_golden.py tests/test_gpu_cloud_cache.py -m gpu -q -x > $O/r06_gputests_call23.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call23.txt; tail -3 $O/r06_gputests_call23.txt
timeout 200 python scripts/sort_bench.py 2>&1 | tail -4 | tee $O/r06_sort_bench_call23.txt
for shape in "64 2" "32 4"; do
  set -- $shape
  for v in base new base new base new; do
    if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
    timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --cpu-baseline 0 --fe-batch $1 --fe-batch-streams $2 > $O/r06_bench_call23_cfg4.json 2> $O/r06_bench_call23_cfg4.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call23_cfg4.json").read().strip().splitlines()[-1])
pk=json.load(open("gpurun_out/bench_detail_cfg4.json"))["per_kernel"]
print("cfg4 $1x$2 $v", {k:d.get(k) for k in ("value_all_pairs","ms_per_step")}, {k:round(pk[k]["ms_total"]/max(1,pk[k]["launches"]),3) for k in ("voxel_sort","fb_grid","fb_voxel","pca_cells","bsc")})
PY
  done
done
unset GHICP_LIB
for v in 0.15 0.25 0.35 0.15 0.25 0.35; do
  t=$RANDOM
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --tail-fraction $v --scene-cache /tmp/scenes64 > $O/r06_bench_call23_tf${v}_$t.json 2> $O/r06_bench_call23_tf${v}_$t.err
  echo "bench tail-fraction $v rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call23_tf${v}_$t.json").read().strip().splitlines()[-1])
print("tf $v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, {k:d["pair_loop_stats"][k] for k in ("longest_solve_ms","mean_launch_span_ms","idle_slot_fraction")}, d.get("batch_ms",{}).get("front_end_ms_per_cloud_on_its_stream"))
PY
done
