#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3h
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err; echo "bench base rc=$?" | tee -a $OUT/summary.txt
MAAI_WGRAD_SIDE_STREAM=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_side.json 2> $OUT/bench_side.err; echo "bench side rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_base2.json 2> $OUT/bench_base2.err; echo "bench base2 rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 400 python3 bench.py --batch 512 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_b512.json 2> $OUT/bench_b512.err; echo "bench b512 rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python3 scripts/eval_bench.py > $OUT/eval_bench.txt 2>&1; echo "eval rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/eval_bench.txt
MAAI_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rehearse_selflaunch.log 2>&1; echo "rehearse rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/rehearse_selflaunch.log | cut -c1-400
MAAI_BENCH_REHEARSE=1 MAAI_P2P_GATHER=1 timeout -k 10 300 python3 bench.py --gpus 2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rehearse_selflaunch_p2p.log 2>&1; echo "rehearse p2p rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/rehearse_selflaunch_p2p.log | cut -c1-300
python3 -c "
import json
for n in ('bench_base','bench_side','bench_base2','bench_b512'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['peak_hbm_GB'], d['config']['peak_hbm_reserved_GB'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
