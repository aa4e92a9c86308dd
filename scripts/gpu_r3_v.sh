#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3y
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 600 python -m pytest tests/test_gpu_c64.py tests/test_gpu_evalfuse.py "tests/test_gpu_timed_size.py::test_batch_split_invariance_at_256_images" -x -q > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/pp_ab.py 256 c64 > $OUT/c64_ab.txt 2>&1; echo "c64_ab rc=$?" | tee -a $OUT/summary.txt; grep -v amdgpu $OUT/c64_ab.txt | cut -c1-150
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c64.json 2> $OUT/bench_c64.err; echo "bench c64 rc=$?" | tee -a $OUT/summary.txt
MAAI_CONV_C64=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_noc64.json 2> $OUT/bench_noc64.err; echo "bench noc64 rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
for n in ('bench_c64','bench_noc64'):
    try:
        d=json.load(open('$OUT/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['config']['loss'])
    except Exception as e: print(n, 'ERR', e)
" | tee -a $OUT/summary.txt
