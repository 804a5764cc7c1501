#!/bin/bash
# Round 3, GPU call 3: Kuhn-Munkres timing after the barrier-free revalidation pass; chip partition sweep (front-end CUs vs solve CUs).
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_frontend.py tests/test_gpu_loop.py tests/test_gpu_batch.py tests/test_gpu_configs.py tests/test_gpu_icp.py tests/test_gpu_cloud_cache.py -m gpu -x -q > $O/r03_gputests_3.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/r03_gputests_3.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg3" >> $O/r03_gputests_3.txt 2>&1
echo "pytest cfg3 rc=$?"; tail -3 $O/r03_gputests_3.txt
(timeout 120 python scripts/km_bench.py --check; GHICP_KM_STATS=1 timeout 120 python scripts/km_bench.py 2>&1 | grep -v "^\[km4 stats\].*$" | head -3; GHICP_KM_STATS=1 timeout 120 python scripts/km_bench.py 2>&1 | grep "km4 stats" | awk 'NR%2==0') > $O/r03_km_bench3.txt 2>&1
cat $O/r03_km_bench3.txt
B="python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes"
for cfg in "0 2" "48 1" "64 1" "32 1" "48 2"; do
  set -- $cfg
  timeout 600 $B --fe-cus $1 --pipeline $2 > $O/r03_split_fe$1_p$2.json 2> $O/r03_split_fe$1_p$2.err
  echo "fe_cus=$1 pipeline=$2 rc=$?"
  python - <<EOF
import json
try:
    d=json.loads(open("$O/r03_split_fe$1_p$2.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["batch_ms"], d["pair_loop_stats"])
except Exception as e:
    print("no line", e); print(open("$O/r03_split_fe$1_p$2.err").read()[-1500:])
EOF
done
