#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3d
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 400 python -m pytest tests/test_gpu_pp.py tests/test_gpu_xf.py -x -q > $OUT/pp_tests.log 2>&1
rc=$?; echo "pp+xf tests rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/pp_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/ppw_ab.py 256 > $OUT/ppw_ab.txt 2>&1; echo "ppw_ab rc=$?" | tee -a $OUT/summary.txt; head -5 $OUT/ppw_ab.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-table $OUT/table.json --detail > $OUT/bench_pp.json 2> $OUT/bench_pp.err; echo "bench pp rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
d=json.load(open('$OUT/bench_pp.json')); print('bench', d['value'], d['ms_per_step'], d['config']['peak_hbm_GB'], d['config']['loss'], d['roofline'])
" | tee -a $OUT/summary.txt
