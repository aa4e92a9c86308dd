#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3e
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
OLD=$PWD/multimodal-active-ai_amd/lib/variants/libmaai_hip_ppold.so
timeout -k 10 400 python -m pytest tests/test_gpu_pp.py -x -q > $OUT/pp_tests.log 2>&1
rc=$?; echo "pp tests rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/pp_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/pp_ab.py 256 c256 > $OUT/pp_ab_new.txt 2>&1; echo "pp_ab new rc=$?" | tee -a $OUT/summary.txt
MAAI_LIB_PATH=$OLD timeout -k 10 300 python scripts/pp_ab.py 256 c256 > $OUT/pp_ab_old.txt 2>&1; echo "pp_ab old rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python scripts/ppw_ab.py 256 > $OUT/ppw_ab_new.txt 2>&1; echo "ppw_ab new rc=$?" | tee -a $OUT/summary.txt
MAAI_LIB_PATH=$OLD timeout -k 10 300 python scripts/ppw_ab.py 256 > $OUT/ppw_ab_old.txt 2>&1; echo "ppw_ab old rc=$?" | tee -a $OUT/summary.txt
paste -d'\n' $OUT/pp_ab_new.txt $OUT/pp_ab_old.txt | grep -v amdgpu | cut -c1-130
paste -d'\n' $OUT/ppw_ab_new.txt $OUT/ppw_ab_old.txt | grep -v amdgpu | cut -c1-130
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_new.json 2> $OUT/bench_new.err; echo "bench new rc=$?" | tee -a $OUT/summary.txt
MAAI_LIB_PATH=$OLD timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_old.json 2> $OUT/bench_old.err; echo "bench old rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
for n in ('bench_new','bench_old'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
