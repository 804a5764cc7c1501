#!/bin/bash
# Round 4, fifth GPU call: schedules.  Fine-grained pipeline (--pipeline 1: a group's loop starts when ITS front ends are done, front ends
# group-major) with 8 / 16 / 32 loop groups against the step-level pipeline (call 4: 395.6 pairs/s), and an earlier front-end start
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
SC="--scene-cache /tmp/scenes64"
run() {
  name=$1; shift
  timeout 420 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 $SC "$@" > $O/r04_sched_$name.json 2> $O/r04_sched_$name.err
  echo "--- $name rc=$?"
  python - $O/r04_sched_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("batch_ms"), d.get("pair_loop_stats"))
except Exception as e: print("parse failed",e)
PY
  tail -2 $O/r04_sched_$name.err | cut -c1-300
}
run p1_g16 --pipeline 1 --loop-groups 16
run p1_g32 --pipeline 1 --loop-groups 32
run p1_g8 --pipeline 1 --loop-groups 8
run p2_tail30 --tail-fraction 0.3
