#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3final
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_driver_flow.py -q -x > $OUT/model_tests.log 2>&1
rc=$?; echo "model tests rc=$rc" | tee -a $OUT/summary.txt; tail -3 $OUT/model_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
bash scripts/profile_round.sh > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/profile_round.log | cut -c1-600
timeout -k 10 600 bash scripts/pmc_conv_shape.sh 64 64 224 3 256 c64_fwd fwd > $OUT/pmc_c64.txt 2>&1; echo "pmc c64 rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/pmc_c64.txt
