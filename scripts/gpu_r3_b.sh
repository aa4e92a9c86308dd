#!/bin/bash
# round 3, call B: the new ping-pong kernel (tests first, under a short limit), its A/B timings, then the rest of the suite
set -o pipefail
OUT=gpurun_out/r3b
mkdir -p $OUT
timeout -k 10 240 python -m pytest tests/test_gpu_pp.py -x -q > $OUT/pp_tests.log 2>&1
rc=$?; echo "pp tests rc=$rc" | tee -a $OUT/summary.txt; tail -15 $OUT/pp_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/pp_ab.py 256 new > $OUT/pp_ab.txt 2>&1; echo "pp_ab rc=$?" | tee -a $OUT/summary.txt; cat $OUT/pp_ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_pp.py > $OUT/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/gputests.log
