#!/bin/bash
# round 3, call C: ping-pong kernels (tests under a short limit), weight-gradient A/B, the whole suite, the bench line
set -o pipefail
OUT=gpurun_out/r3c
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 300 python -m pytest tests/test_gpu_pp.py -x -q > $OUT/pp_tests.log 2>&1
rc=$?; echo "pp tests rc=$rc" | tee -a $OUT/summary.txt; tail -12 $OUT/pp_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/ppw_ab.py 256 > $OUT/ppw_ab.txt 2>&1; echo "ppw_ab rc=$?" | tee -a $OUT/summary.txt; cat $OUT/ppw_ab.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=6 --deselect tests/test_gpu_pp.py > $OUT/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/gputests.log | cut -c1-300
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_pp.json 2> $OUT/bench_pp.err; echo "bench pp rc=$?" | tee -a $OUT/summary.txt
MAAI_CONV_PP=0 MAAI_WGRAD_PP=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_nopp.json 2> $OUT/bench_nopp.err; echo "bench nopp rc=$?" | tee -a $OUT/summary.txt
MAAI_WGRAD_PP=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_ppfwd.json 2> $OUT/bench_ppfwd.err; echo "bench ppfwd rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
for n in ('bench_pp','bench_nopp','bench_ppfwd'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['peak_hbm_GB'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
