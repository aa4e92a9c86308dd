#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3n
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
MAAI_CONV_C64=0 timeout -k 10 400 python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --detail --profile-table $OUT/table.json > $OUT/bench_detail.json 2> $OUT/bench_detail.err; echo "bench detail rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 600 bash scripts/pmc_conv_shape.sh 64 64 224 3 256 c64_fwd fwd > $OUT/pmc_c64.txt 2>&1; echo "pmc c64 rc=$?" | tee -a $OUT/summary.txt
tail -20 $OUT/pmc_c64.txt
