#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3z
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail 5 > $OUT/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" | tee -a $OUT/summary.txt; tail -6 $OUT/gpu_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/smoke.log
timeout -k 10 400 python3 bench.py --batch 512 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_b512.json 2> $OUT/bench_b512.err; echo "bench512 rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
d=json.load(open('$OUT/bench_b512.json')); print('b512', d['value'], d['ms_per_step'])
" | tee -a $OUT/summary.txt
