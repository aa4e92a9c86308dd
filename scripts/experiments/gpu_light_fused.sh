#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3al
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_timed_size.py::test_recompute_step_at_512_images_per_gpu tests/test_gpu_dist.py -q --maxfail 5 -k "recompute or two_ranks or bf16_production" > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit 1; fi
for v in 256 0 256 0; do
MAAI_FUSE_NOGRAD_MAX_CIN=$v timeout -k 10 400 python3 bench.py --batch 512 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_b512_$v.json 2> $OUT/bench_b512_$v.err || { tail -3 $OUT/bench_b512_$v.err; exit 1; }
python3 -c "
import json
d=json.load(open('$OUT/bench_b512_$v.json')); print('b512 nograd_max_cin $v', d['value'], d['ms_per_step'])
" | tee -a $OUT/summary.txt
done
