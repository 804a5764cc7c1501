#!/bin/bash
# Round 4, after the final call: the front end alone on one stream (uncontended kernel times) for four builds of the BSC kernel --
# HEAD (contract expf, exact depth sums through a wrapping 64-bit word + carry, ring), the same without the ring, the same with the
# device library's exp, and round 3's kernel (f64 atomics in arrival order, library exp, no ring)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for v in head noring ocml r03; do
  lib=$R/gh-icp_amd/libghicp_var_$v.so; [ $v = head ] && lib=$R/gh-icp_amd/libghicp_hip.so
  B1="python $R/bench.py --steps 1 --warmup 1 --distinct 8 --pairs-per-step 256 --cpu-baseline 0 --fe-batch 32 --fe-batch-streams 1 --fe-streams 1 --pipeline 0 --scene-cache /tmp/scenes64"
  GHICP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o a -- $B1 > /dev/null 2> $O/r04_bscvar_$v.err
  python $R/scripts/rocprof_summary.py /tmp/prof_$v $O/r04_bscvar_$v.txt "BSC variant $v: $B1" > /dev/null
  echo "--- $v"; grep -h "k_fb_bsc\|k_fb_pca_cells" $O/r04_bscvar_$v.txt | cut -c1-150
done
