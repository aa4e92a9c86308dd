#!/bin/bash
set -o pipefail
OUT=gpurun_out/p3; mkdir -p $OUT
timeout -k 10 400 python3 -m pytest tests/test_gpu_fold.py -m gpu -q -s > $OUT/fold.log 2>&1; echo "fold rc=$?" | tee $OUT/summary.txt
grep -E "folded|passed|failed|Error|assert" $OUT/fold.log | tail -60
timeout -k 10 900 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_fold.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/pytest.log
for cfg in "1 1" "0 0"; do
  set -- $cfg
  MAAI_FOLD=$1 MAAI_FOLD_FWD=$2 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fold$1$2.json 2> $OUT/bench_fold$1$2.err; echo "bench fold=$1 fwd=$2 rc=$?" | tee -a $OUT/summary.txt
  tail -2 $OUT/bench_fold$1$2.err
  python3 -c "import json;d=json.load(open('$OUT/bench_fold$1$2.json'));print('fold=$1 fwd=$2',d['value'],d['ms_per_step'],d['config']['peak_hbm_GB'],d['config']['loss'])"
done
MAAI_FOLD=1 timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --detail --profile-table $OUT/table.json > $OUT/bench_detail.json 2> $OUT/bench_detail.err; echo "detail rc=$?" | tee -a $OUT/summary.txt
