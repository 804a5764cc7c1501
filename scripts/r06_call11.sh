#!/bin/bash
# Round 6, call 11: the confined class's slots go on with the other class's queue when their own is dry (no re-launch behind the kernel).
# Loop tests, default bench base / new (twice).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_loop.py tests/test_golden.py tests/test_gpu_batch.py tests/test_gpu_zz_batch_fullsize.py -m gpu -q -x > $O/r06_gputests_call11.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call11.txt; tail -3 $O/r06_gputests_call11.txt
for v in prev new prev2 new2; do
  if [ ${v:0:4} = prev ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_prev.so; else unset GHICP_LIB; fi
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call11_$v.json 2> $O/r06_bench_call11_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call11_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call11_$v.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"))
t=json.load(open("gpurun_out/r06_bench_call11_${v}_detail.json"))["timeline"]
print("   ", [b["span_s"] for b in t["last_batches"]], t["last_batches"][-1]["active_pairs_every_250ms"])
PY
done
