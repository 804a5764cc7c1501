#!/bin/bash
# Round 5: what bounds the two big front-end kernels?  Counter passes (counters only, one group per run) over the front end alone on one
# stream (16 batches of 32 clouds): issue / wait split, VALU and LDS instruction mix, LDS bank conflicts, fabric traffic.
#   gpurun --timeout 600 -- 'bash scripts/r05_fe_pmc.sh'   -> gpurun_out/r05_fe_pmc_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
B="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --no-hints-steps 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
for G in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $G | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/prof_fe_$T -o p -- $B > /dev/null 2> $O/r05_fe_pmc_$T.err
  python $R/scripts/rocprof_summary.py /tmp/prof_fe_$T $O/r05_fe_pmc_$T.txt "pmc $G: front end on one stream, $B" > /dev/null
  grep -h "k_fb_pca_cells\|k_fb_bsc \|k_fb_bsc$\|k_fb_nms_greedy\|k_cell_start_fill" $O/r05_fe_pmc_$T.txt | cut -c1-200
done
