#!/bin/bash
# Round 6, call 22: the radix sort's small paths (k_rs_small: one launch up to one tile; k_rs_chunk_bases: three launches per place up to 64 tiles) on the MI355X:
# front-end tests, the sort's timing, the reference's call sequence through the drop-in headers (cloud by cloud), cfg4 (front-end bound) base / new.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_batch.py tests/test_golden.py tests/test_gpu_cloud_cache.py tests/test_gpu_icp.py tests/test_gpu_dropin.py -m gpu -q -x > $O/r06_gputests_call22.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call22.txt; tail -3 $O/r06_gputests_call22.txt
timeout 200 python scripts/sort_bench.py 2>&1 | tail -4 | tee $O/r06_sort_bench_call22.txt
timeout 300 python scripts/dropin_time.py > $O/r06_dropin_time_call22.out 2> $O/r06_dropin_time_call22.err; echo "dropin rc=$?"; tail -c 700 $O/r06_dropin_time_call22.out; echo   # (links the in-tree library; call 20 has the library before)
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 300 python bench.py --config 4 --steps 4 --warmup 1 --cpu-baseline 0 > $O/r06_bench_call22_cfg4_$v.json 2> $O/r06_bench_call22_cfg4_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call22_cfg4_$v.json").read().strip().splitlines()[-1])
print("cfg4 $v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("batch_ms",{}).get("front_end_calibration"))
PY
done
