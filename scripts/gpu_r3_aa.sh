#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3aa
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 600 python -m pytest tests/test_gpu_c64.py -x -q > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $OUT/summary.txt; tail -3 $OUT/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/pp_ab.py 256 c64 > $OUT/c64_ab.txt 2>&1; echo "c64_ab rc=$?" | tee -a $OUT/summary.txt; grep -v amdgpu $OUT/c64_ab.txt | cut -c1-150
timeout -k 10 600 bash scripts/pmc_conv_shape.sh 64 64 224 3 256 c64_fwd fwd > $OUT/pmc_c64.txt 2>&1; echo "pmc c64 rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/pmc_c64.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
d=json.load(open('$OUT/bench.json')); print('bench', d['value'], d['ms_per_step'])
" | tee -a $OUT/summary.txt
