#!/bin/bash
# Round 5, sixth GPU call (~6 GPU-minutes): the three-per-CU Kuhn-Munkres class confined to its own CUs (LDS fragmentation), A/B on one box.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_loop.py -m gpu -x -q > $O/r05_gputests_call6.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r05_gputests_call6.txt; tail -3 $O/r05_gputests_call6.txt
for C in 0 1; do
  GHICP_LOOP_CONFINE=$C timeout 400 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r05_bench_confine$C.json 2> $O/r05_bench_confine$C.err
  echo "bench confine=$C rc=$?"; cp $O/bench_detail_cfg2.json $O/r05_bench_confine${C}_detail.json; python - $C <<'PY'
import json,sys
C=sys.argv[1]
d=json.loads(open("gpurun_out/r05_bench_confine%s.json"%C).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"))
t=json.load(open("gpurun_out/r05_bench_confine%s_detail.json"%C))["timeline"]
print(t["loop_calls_s"]); print([ (b["span_s"], b["active_pairs_every_250ms"]) for b in t["last_batches"]])
PY
done
