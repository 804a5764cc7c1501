#!/bin/bash
# First GPU call of the next round: schedule sweeps of bench.py with the batched front end on a reduced cfg2 job
# (16 distinct scenes, 1344 pairs per step) so that each point costs ~1 GPU-minute, then a kernel trace of the batched front end.
#   gpurun --timeout 1500 -- 'bash scripts/r03_sweeps.sh'
# Every run writes its JSON line to gpurun_out/r03_sweep_<name>.json; summary at the end.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
COMMON="--config 2 --distinct 16 --pairs-per-step 1344 --steps 2 --warmup 1 --cpu-baseline 0"
run() {  # name, extra args
  local name=$1; shift
  timeout 170 python bench.py $COMMON "$@" > gpurun_out/r03_sweep_$name.json 2> gpurun_out/r03_sweep_$name.err || echo "$name: rc=$?"
}
run fe_cloud_by_cloud --fe-batch 0
run fe_batch32_s4 --fe-batch 32 --fe-streams 4
run fe_batch64_s2 --fe-batch 64 --fe-streams 2
run fe_batch16_s8 --fe-batch 16 --fe-streams 8
run tail40 --fe-batch 32 --fe-streams 4 --tail-fraction 0.4
run tail60 --fe-batch 32 --fe-streams 4 --tail-fraction 0.6
run pipeline1 --fe-batch 32 --fe-streams 4 --pipeline 1
run groups6 --fe-batch 32 --fe-streams 4 --loop-groups 6
GHICP_PCA_CHUNK=256 run pca_chunk256 --fe-batch 32 --fe-streams 4   # 4 KB PCA tile: 8 instead of 5 waves per SIMD (same results)
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r03_sweep_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-28s %8.1f pairs/s  %9.1f ms/step  fe %s" % (f.split("r03_sweep_")[1][:-5], d["value"], d["ms_per_step"], d["batch_ms"].get("front_end_calibration") or d["config"].get("fe_batch")))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
PY
# kernel trace of the batched front end alone (cfg4: the loop is 4 % of a step there)
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/r03_prof_fe" -- python "$OLDPWD/bench.py" --config 4 --steps 3 --warmup 1 --cpu-baseline 0 --fe-batch 32 --fe-streams 4 > "$OLDPWD/gpurun_out/r03_prof_fe.log" 2>&1
cd "$OLDPWD"
f=$(find gpurun_out/r03_prof_fe -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -25 "$f" > gpurun_out/r03_prof_fe_kernel_stats_head.csv && cat gpurun_out/r03_prof_fe_kernel_stats_head.csv
