#!/bin/bash
# One GPU call's worth of evidence for a round (run ON THE GPU BOX from the repo root via gpurun): bench lines, phase timings,
# decode / host / single-frame side measurements, then the rocprofv3 passes of profiles/collect.sh and tools/sq_stage.sh.
# Every step under its own timeout; a step that fails or is killed ends the script (no GPU step is started behind it).
set -e
TAG=${1:-r4a}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench c3 done"
timeout -k 10 300 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err
timeout -k 10 300 python bench.py --channels 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_bgr.json 2> $OUT/${TAG}_bench_bgr.err
timeout -k 10 300 python bench.py --workload c1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_c1.json 2> $OUT/${TAG}_bench_c1.err
timeout -k 10 300 python bench.py --workload real --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_real.json 2> $OUT/${TAG}_bench_real.err
echo "bench c5 / bgr / c1 / real done"
timeout -k 10 300 python tools/gpu_stage_phase.py 1536 c3 > $OUT/${TAG}_stage_phase_timing_c3.log 2>&1
timeout -k 10 300 python tools/gpu_ncc_phase.py 512 > $OUT/${TAG}_ncc_phase_timing.log 2>&1
timeout -k 10 120 python tools/gpu_single_frame.py > $OUT/${TAG}_single_frame.log 2>&1
timeout -k 10 300 python tools/gpu_decode_path.py 2048 > $OUT/${TAG}_decode_path.log 2>&1
echo "phases / single frame / decode done"
timeout -k 10 900 bash profiles/collect.sh $TAG 1536 > $OUT/collect_$TAG.log 2>&1
echo "collect done"
timeout -k 10 400 bash tools/sq_stage.sh $TAG 1536 > $OUT/sq_stage_$TAG.log 2>&1
echo "all done"
