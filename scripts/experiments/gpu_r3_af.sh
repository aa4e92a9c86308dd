#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3ag
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 240 python -m pytest tests/test_gpu_pp.py tests/test_gpu_evalfuse.py -q --maxfail 6 > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $OUT/summary.txt; tail -8 $OUT/tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit 1; fi
for k in 1 0 1 0; do
  echo "== ppp $k" >> $OUT/ab.txt
  MAAI_CONV_PPP=$k timeout -k 10 300 python scripts/pp_ab.py 256 c256 2>&1 | grep -v amdgpu | cut -c1-130 >> $OUT/ab.txt || exit 1
done
MAAI_CONV_PPP=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_d1.json 2> $OUT/bench_d1.err; echo "bench d1 rc=$?" | tee -a $OUT/summary.txt
MAAI_CONV_PPP=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_d0.json 2> $OUT/bench_d0.err; echo "bench d0 rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
for n in ('bench_d1','bench_d0'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
