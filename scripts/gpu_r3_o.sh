#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3o
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
for epf in 1 0 2; do
MAAI_C64_EPF=$epf timeout -k 10 300 python -m pytest tests/test_gpu_c64.py -x -q > $OUT/c64_tests_$epf.log 2>&1
rc=$?; echo "c64 tests epf=$epf rc=$rc" | tee -a $OUT/summary.txt; tail -3 $OUT/c64_tests_$epf.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
done
for epf in 1 0 2; do
MAAI_C64_EPF=$epf timeout -k 10 300 python scripts/pp_ab.py 256 c64 > $OUT/c64_ab_$epf.txt 2>&1; echo "c64_ab epf=$epf rc=$?" | tee -a $OUT/summary.txt; grep -v amdgpu $OUT/c64_ab_$epf.txt | cut -c1-150
done
