#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3i
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
MAAI_WGRAD_SIDE_STREAM=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_side.json 2> $OUT/bench_side.err; echo "bench side rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err; echo "bench base rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
for n in ('bench_side','bench_base'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
unset MAAI_WGRAD_TUNE_FILE
bash scripts/profile_round.sh > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/profile_round.log | cut -c1-600
