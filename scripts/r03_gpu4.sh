#!/bin/bash
# Round 3, GPU call 4: A/B of the serial-wave spreading (GHICP_KM_SPREAD) on a short bench; DFS iteration kinds with cycles.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
GHICP_KM_STATS=1 timeout 120 python scripts/km_bench.py 2>&1 | grep "km4 dfs" | awk 'NR%2==0' > $O/r03_km_dfs_kinds.txt
cat $O/r03_km_dfs_kinds.txt
B="python bench.py --steps 2 --warmup 1 --distinct 16 --pairs-per-step 2048 --pipeline 0 --cpu-baseline 0 --scene-cache /tmp/scenes"
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export GHICP_KM_SPREAD=1; else unset GHICP_KM_SPREAD; fi
  timeout 400 $B > $O/r03_spread$v.json 2> $O/r03_spread$v.err
  python - <<EOF
import json
try:
    d=json.loads(open("$O/r03_spread$v.json").read().strip().splitlines()[-1])
    print("spread=$v", {k:d[k] for k in ("value","ms_per_step","ms_per_iteration")}, d["pair_loop_stats"])
except Exception as e:
    print("no line", e); print(open("$O/r03_spread$v.err").read()[-1500:])
EOF
done
