#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3g
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py -x -q -k "direct_allgather" > $OUT/comm_tests.log 2>&1
echo "comm tests rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/comm_tests.log | cut -c1-400
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=8 --deselect tests/test_gpu_dist.py::test_direct_allgather_matches_process_group_gather --deselect tests/test_gpu_dist.py::test_two_ranks_direct_allgather_transport > $OUT/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/gputests.log | cut -c1-300
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
d=json.load(open('$OUT/bench.json')); print('bench', d['value'], d['ms_per_step'], d['config']['peak_hbm_GB'], d['config']['loss'])
" | tee -a $OUT/summary.txt
