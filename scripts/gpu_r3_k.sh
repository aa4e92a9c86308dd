#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3k
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
V=$PWD/multimodal-active-ai_amd/lib/variants
for v in base exp8 exp16 base2 exp8b exp16b; do
  case $v in base|base2) unset MAAI_LIB_PATH;; exp8|exp8b) export MAAI_LIB_PATH=$V/libmaai_hip_exp8.so;; exp16|exp16b) export MAAI_LIB_PATH=$V/libmaai_hip_exp16.so;; esac
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "$v rc=$?" | tee -a $OUT/summary.txt
done
python3 -c "
import json
for n in ('base','exp8','exp16','base2','exp8b','exp16b'):
    try:
        d=json.load(open('$OUT/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
