#!/bin/bash
OUT=gpurun_out/r3ai
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
for v in 0 256 0 256; do
MAAI_FUSE_NOGRAD_MAX_CIN=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err || { tail -5 $OUT/bench_$v.err; exit 1; }
python3 -c "
import json
d=json.load(open('$OUT/bench_$v.json')); print('nograd_max_cin $v', d['value'], d['ms_per_step'], d['config']['loss'])
" | tee -a $OUT/summary.txt
done
MAAI_FUSE_NOGRAD_MAX_CIN=256 timeout -k 10 400 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3 | cut -c1-200 | tee -a $OUT/summary.txt
