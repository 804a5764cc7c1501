#!/bin/bash
# Round 6, call 8: lazy S in the Kuhn-Munkres solver (rule R5': the search starts without S, S is computed when it first steps back).
# Five real matrices new / base, the loop tests, default bench base / new.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_loop.py tests/test_golden.py tests/test_gpu_batch.py tests/test_gpu_zz_batch_fullsize.py -m gpu -q -x > $O/r06_gputests_call8.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call8.txt; tail -3 $O/r06_gputests_call8.txt
for v in new base new base; do
  L=""; [ $v = base ] && L="--lib $R/gh-icp_amd/libghicp_var_base.so"
  echo "== km_bench $v" | tee -a $O/r06_km_variants_call8.txt; timeout 200 python scripts/km_bench.py --more --check $L 2>&1 | grep "^it" | tee -a $O/r06_km_variants_call8.txt
done
GHICP_KM_STATS=1 timeout 200 python scripts/km_bench.py --more 2>&1 | grep "km4 stats\|km4 dfs\|^it" > $O/r06_km4_stages.txt; head -4 $O/r06_km4_stages.txt | cut -c1-400
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call8_$v.json 2> $O/r06_bench_call8_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call8_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call8_$v.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
PY
done
