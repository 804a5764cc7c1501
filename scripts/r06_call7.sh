#!/bin/bash
# Round 6, call 7: NMS candidates in rank order inside their cells (a walk leaves a cell at the first entry of lower rank).
# Tests, front end on one stream, default bench base / new.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_frontend.py tests/test_golden.py tests/test_gpu_cloud_cache.py tests/test_gpu_zz_batch_fullsize.py -m gpu -q -x > $O/r06_gputests_call7.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call7.txt; tail -3 $O/r06_gputests_call7.txt
cd /tmp
B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_new -o a -- $B1 > /dev/null 2> $O/r06_fe_call7_new.err
python $R/scripts/rocprof_summary.py /tmp/prof_new $O/r06_kernel_stats_fe_one_stream_call7_new.txt "front end on one stream (call 7, new): $B1" | head -30 | cut -c1-150
python - <<PY
import csv, glob
rows=[]
for f in glob.glob("/tmp/prof_new/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fb_nmsr_round" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
rows.sort()
print("k_fb_nmsr_round durations (us), first three batches:", [round(d,1) for _,d in rows[:42]])
PY
cd $R
for v in base new; do
  if [ $v = base ]; then export GHICP_LIB=$R/gh-icp_amd/libghicp_var_base.so; else unset GHICP_LIB; fi
  timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --no-hints-steps 0 --scene-cache /tmp/scenes64 > $O/r06_bench_call7_$v.json 2> $O/r06_bench_call7_$v.err
  echo "bench $v rc=$?"; cp $O/bench_detail_cfg2.json $O/r06_bench_call7_${v}_detail.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_call7_$v.json").read().strip().splitlines()[-1])
print("$v", {k:d.get(k) for k in ("value","value_all_pairs","ms_per_step")}, d.get("pair_loop_stats"), d.get("batch_ms"))
t=json.load(open("gpurun_out/r06_bench_call7_${v}_detail.json"))
print(t["front_end_calibration"])
PY
done
