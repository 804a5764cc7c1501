#!/bin/bash
# last GPU call of round 3: sanity of the final host-side edits (per-batch launch records, progress reset) -- loop tests + a short bench
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_loop.py tests/test_gpu_cloud_cache.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 2 --warmup 1 --distinct 16 --pairs-per-step 2048 --cpu-baseline 0 > $O/r03_bench_sanity.json 2> $O/r03_bench_sanity.err
echo "bench rc=$?"; tail -c 2600 $O/r03_bench_sanity.json
