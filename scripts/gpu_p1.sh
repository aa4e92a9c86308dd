#!/bin/bash
set -o pipefail
OUT=gpurun_out/p1; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_fold.py -m gpu -x -q -s > $OUT/fold.log 2>&1; echo "fold rc=$?" | tee $OUT/summary.txt
grep -E "folded|passed|failed|Error" $OUT/fold.log | tail -30
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_fold.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest.log
for f in 1 0; do
  MAAI_FOLD=$f timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fold$f.json 2> $OUT/bench_fold$f.err; echo "bench fold=$f rc=$?" | tee -a $OUT/summary.txt
  python3 -c "import json;d=json.load(open('$OUT/bench_fold$f.json'));print('fold=$f',d['value'],d['ms_per_step'],d['config']['peak_hbm_GB'],d['config']['loss'])"
done
