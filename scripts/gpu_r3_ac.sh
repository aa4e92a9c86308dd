#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3ac
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_evalfuse.py -q --maxfail 5 > $OUT/evalfuse.log 2>&1
rc=$?; echo "evalfuse rc=$rc" | tee -a $OUT/summary.txt; tail -4 $OUT/evalfuse.log | cut -c1-250
MAAI_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rehearse_selflaunch.log 2>&1; echo "rehearse rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/rehearse_selflaunch.log | cut -c1-400
MAAI_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --batch 64 --recompute --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rehearse_recompute.log 2>&1; echo "rehearse recompute rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/rehearse_recompute.log | cut -c1-400
