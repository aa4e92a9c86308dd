#!/bin/bash
# round 3, call A: the whole GPU suite, the bench line, the drain A/B, the self-launched 2-rank rehearsal
set -o pipefail
OUT=gpurun_out/r3a
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/gputests.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_new.json 2> $OUT/bench_new.err; echo "bench new rc=$?" | tee -a $OUT/summary.txt
MAAI_LIB_PATH=$PWD/multimodal-active-ai_amd/lib/variants/libmaai_hip_nodrain.so timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_nodrain.json 2> $OUT/bench_nodrain.err; echo "bench nodrain rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_new2.json 2> $OUT/bench_new2.err; echo "bench new2 rc=$?" | tee -a $OUT/summary.txt
MAAI_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/rehearse_selflaunch.log 2>&1; echo "rehearse rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
for n in ('bench_new','bench_nodrain','bench_new2'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['peak_hbm_GB'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
tail -3 $OUT/rehearse_selflaunch.log
