#!/bin/bash
# Round 5, first GPU call (after scripts/r05_prepare.sh; ~27 GPU-minutes): every staged candidate of round 4 timed against the shipped library ON ONE BOX
# (box-to-box noise of bench.py is +-3 %, so only numbers of the same call are compared).
#   gpurun --timeout 1900 -- 'bash scripts/r05_call1.sh'
# Writes gpurun_out/r05_*: copy what is to be judged into profiles/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
L=$R/gh-icp_amd
mkdir -p $O
export TMPDIR=/tmp
# (0) the wave barrier added to k4_bulk after round 4's last GPU call (a scheduling fence, no instruction): the solver tests first
timeout 300 python -m pytest tests/test_gpu_loop.py -m gpu -x -q > $O/r05_gputests_km_loop.txt 2>&1
echo "pytest km+loop rc=$?" | tee -a $O/r05_gputests_km_loop.txt; tail -2 $O/r05_gputests_km_loop.txt
# (1) single solves of the five real matrices: shipped against the fused S rounds
{
for v in main sfused; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  echo "--- $v"
  timeout 200 python scripts/km_bench.py --lib $lib --more --check 2>&1 | grep solve
done
} > $O/r05_km_variants.txt 2>&1
cat $O/r05_km_variants.txt
# (2) do small kernels make progress beside four resident slots?  shipped (128 VGPRs) against the 96-VGPR loop
for v in main occ5; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  GHICP_LIB=$lib timeout 240 python scripts/r05_coresidency_probe.py > $O/r05_probe_$v.json 2> $O/r05_probe_$v.err
  echo "--- probe $v rc=$?"; tail -1 $O/r05_probe_$v.json | cut -c1-600
done
# (3) the bench line per library, default schedule (front ends of step k+1 start in the TAIL of step k): what each change costs or gives by itself
show() {
  python - $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), (d.get("batch_ms") or {}).get("front_end_ms_per_cloud_on_its_stream"))
except Exception as e: print("parse failed",e)
PY
}
B="--steps 2 --warmup 1 --cpu-baseline 0 --scene-cache /tmp/scenes64"
for v in main beside; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  GHICP_LIB=$lib timeout 300 python bench.py $B > $O/r05_bench_var_$v.json 2> $O/r05_bench_var_$v.err
  echo "--- bench $v rc=$?"; show $O/r05_bench_var_$v.json
done
# (3b) THE experiment of lever 1: front ends of step k+1 issued while step k is dense (--tail-fraction 1.0: they start with the loop).  With the shipped
#      library they wait for the tail anyway (same-box baseline; round 4 measured this schedule as slightly worse); with `beside` they can back-fill;
#      `besides` (its own tree under variants/: scripts/r05_prepare.sh) adds the small-LDS primitives, always (1) or only while the loops are dense (auto)
for v in main beside; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  GHICP_LIB=$lib timeout 300 python bench.py $B --tail-fraction 1.0 > $O/r05_bench_early_$v.json 2> $O/r05_bench_early_$v.err
  echo "--- bench early front ends, $v rc=$?"; show $O/r05_bench_early_$v.json
done
for m in 1 auto; do
  ( cd $R/variants/besides && timeout 300 python bench.py $B --tail-fraction 1.0 --fe-small-lds $m > $O/r05_bench_early_besides_small_$m.json 2> $O/r05_bench_early_besides_small_$m.err )
  echo "--- bench early front ends, besides small=$m rc=$?"; show $O/r05_bench_early_besides_small_$m.json
done
( cd $R/variants/besides && GHICP_LOOP_HI_PRIO=1 timeout 300 python bench.py $B --tail-fraction 1.0 --fe-small-lds auto > $O/r05_bench_early_besides_auto_hiprio.json 2> $O/r05_bench_early_besides_auto_hiprio.err )
echo "--- bench early front ends, besides small=auto, loop streams at the highest priority rc=$?"; show $O/r05_bench_early_besides_auto_hiprio.json
# (4) front end alone on one stream: shipped against the packed voxel sort (the Onesweep rows of the two summaries)
for v in main packed; do
  lib=$L/libghicp_var_$v.so; [ $v = main ] && lib=$L/libghicp_hip.so
  GHICP_LIB=$lib bash scripts/r04_fe_check.sh > $O/r05_fe_check_$v.log 2>&1
  cp $O/r04_fe_check.txt $O/r05_fe_one_stream_$v.txt 2>/dev/null
  grep -h "radix\|onesweep\|Onesweep" $O/r05_fe_one_stream_$v.txt | cut -c1-150 | head -6
done
# (5) Infinity-Cache knee of the solve slots (scripts/r05_slots_knee.sh), last: it is the cheapest to lose
bash $R/scripts/r05_slots_knee.sh
