#!/bin/bash
# Round 5 (prepared at the end of round 4, never run): mean Kuhn-Munkres solve time against the number of busy slots -- does the 1.18 x between one slot
# per two CUs (128 pairs in a step) and four per CU (1024) come in one step where the slots' CSR stops fitting the 256 MB Infinity Cache (~1 MB per slot:
# a knee between 256 and 512) or gradually (a per-CU resource)?  Same 8 scenes, no profiler; ~2 GPU-minutes.
#   gpurun --timeout 400 -- 'bash scripts/r05_slots_knee.sh'   -> gpurun_out/r05_slots_knee.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
: > $O/r05_slots_knee.txt
for P in 128 256 384 512 768 1024; do
  timeout 120 python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step $P --cpu-baseline 0 --pipeline 0 --tail-fraction 0 --scene-cache /tmp/scenes \
    > $O/r05_knee_p$P.json 2> $O/r05_knee_p$P.err
  echo "P=$P $(grep -h -o '"pair_loop_stats": {[^}]*}' $O/r05_knee_p$P.json)" | tee -a $O/r05_slots_knee.txt
done
