#!/bin/bash
# Round 4, second GPU call: the new paths on the MI355X (int8-MFMA Hamming distance over a batch, ghicp_iterate, batched final transform,
# drop-in main() from raw clouds with the FPFH + NNR branch, seeded flood merged), then the bench with the slot timeline, the same with a
# strict front-end / loop schedule (--pipeline 0: what the two parts cost on their own), and a kernel trace.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_dropin.py tests/test_gpu_cloud_cache.py tests/test_gpu_batch.py tests/test_gpu_configs.py -m gpu -x -q --durations=5 > $O/r04_gputests_call2.txt 2>&1
echo "pytest rc=$?"; tail -12 $O/r04_gputests_call2.txt
SC="--scene-cache /tmp/scenes64"
timeout 500 python bench.py --steps 3 --warmup 1 --cpu-baseline 0 $SC > $O/r04_bench_call2.json 2> $O/r04_bench_call2.err
echo "bench rc=$?"; tail -c 3000 $O/r04_bench_call2.json; cp $O/bench_detail_cfg2.json $O/r04_bench_call2_detail.json
timeout 500 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --pipeline 0 $SC > $O/r04_bench_call2_p0.json 2> $O/r04_bench_call2_p0.err
echo "bench p0 rc=$?"; tail -c 1500 $O/r04_bench_call2_p0.json; cp $O/bench_detail_cfg2.json $O/r04_bench_call2_p0_detail.json
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 $SC > $O/r04_bench_under_rocprof.json 2> $O/r04_rocprof.err
python $R/scripts/rocprof_summary.py /tmp/prof_kt $O/r04_kernel_stats_bench_default.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-baseline 0" | head -30
python $R/scripts/rocprof_timeline.py /tmp/prof_kt $O/r04_timeline_bench_default.txt; head -3 $O/r04_timeline_bench_default.txt
