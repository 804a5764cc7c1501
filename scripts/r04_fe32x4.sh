#!/bin/bash
# last GPU seconds of round 4: the default schedule with the front end fixed at 32 clouds x 4 streams (the calibration usually picks 16 x 8)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 150 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 4 --scene-cache /tmp/scenes64 > $O/r04_bench_fe32x4.json 2> $O/r04_bench_fe32x4.err
echo "rc=$?"; python - <<'PY'
import json,collections
d=json.loads(open('/root/repo/gpurun_out/r04_bench_fe32x4.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d["batch_ms"], d["pair_loop_stats"])
t=json.load(open('/root/repo/gpurun_out/bench_detail_cfg2.json'))["timeline"]
print("loop calls", t["loop_calls_s"])
busy=collections.Counter()
for a,b,n in t["fe_calls_s"]:
    x=a
    while x<b:
        i=int(x); nx=min(b,i+1); busy[i]+=nx-x; x=nx
print("fe busy", [round(busy[i],1) for i in range(int(max(b for a,b,n in t["fe_calls_s"]))+1)])
PY
