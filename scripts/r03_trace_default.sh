#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench command (fewer steps), final round-3 build: the k_pair_loop dispatches beside bench.py's own
# HIP-event time of the same run (roofline.avg_launch_ms).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -o d -- $B > $O/r03_bench_default_under_rocprof.json 2> $O/r03_rocprof_d.err
python $R/scripts/rocprof_summary.py /tmp/prof_d $O/r03_kernel_stats_bench_default.txt "$B" > /dev/null
head -16 $O/r03_kernel_stats_bench_default.txt | cut -c1-150
python - <<EOF
import json
d=json.loads(open("$O/r03_bench_default_under_rocprof.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["pair_loop_stats"])
EOF
