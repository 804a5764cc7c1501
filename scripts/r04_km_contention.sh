#!/bin/bash
# NEXT ROUND (prepared at the end of round 3, never run): why is a Kuhn-Munkres solve 1.6 x slower with four slots per CU than alone?
# Same persistent pair loop, same 8 scenes, once with ~1 busy slot per CU (128 pairs in a step) and once with every slot busy (1024):
# PMC passes (counters only, one group per run as the guide prescribes) for the instruction cache, the issue/wait split and the LDS.
#   gpurun --timeout 1500 -- 'bash scripts/r04_km_contention.sh'   -> gpurun_out/r04_contention_*.txt ; copy the summaries to profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
SC="--scene-cache /tmp/scenes"
for P in 128 1024; do
  B="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step $P --cpu-baseline 0 --pipeline 0 --tail-fraction 0 $SC"
  timeout 300 $B > $O/r04_contention_p${P}_bench.json 2> $O/r04_contention_p${P}_bench.err   # pair_loop_stats.mean_solve_ms without a profiler
  for G in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    T=$(echo $G | tr ' ' '_')
    timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/prof_${P}_$T -o p -- $B > /dev/null 2> $O/r04_contention_p${P}_$T.err
    python $R/scripts/rocprof_summary.py /tmp/prof_${P}_$T $O/r04_contention_p${P}_$T.txt "pmc $G: $B" > /dev/null
    grep -h "k_pair_loop" $O/r04_contention_p${P}_$T.txt | cut -c1-220
  done
done
grep -h -o '"pair_loop_stats": {[^}]*}' $O/r04_contention_p128_bench.json $O/r04_contention_p1024_bench.json
