#!/bin/bash
# Round 4, first GPU call: the four staged Kuhn-Munkres candidates of round 3 timed against main
#  (1) single solves on the five real matrices, checked against the oracle; (2) bench.py --steps 3 per library; (3) contention PMC passes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
L=$R/gh-icp_amd
{
for v in main seeded five slack all3; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  echo "--- $v"
  timeout 200 python scripts/km_bench.py --lib $lib --more --check 2>&1 | grep solve
done
} > $O/r04_km_variants.txt 2>&1
cat $O/r04_km_variants.txt
for v in main five slack all3 seeded; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  GHICP_LIB=$lib timeout 400 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes64 > $O/r04_bench_var_$v.json 2> $O/r04_bench_var_$v.err
  echo "--- bench $v rc=$?"
  python - $O/r04_bench_var_$v.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step")}, d.get("pair_loop_stats"))
except Exception as e: print("parse failed",e)
PY
done
timeout 1200 bash scripts/r04_km_contention.sh 2>&1 | tail -40
