#!/bin/bash
# One GPU-box call that refreshes everything profiles/ holds for a round (rNN = $1): the bench line with the PMC traffic in
# it, the rocprofv3 kernel-trace summary and PMC passes of the same command (scripts/profile_round.sh), the per-shape table,
# the 512-image lines, the eval-mode forward.   gpurun -- 'PART=1 bash scripts/gpu_round_profiles.sh r04', then PART=2
set -o pipefail
R=${1:-r04}
OUT=gpurun_out/$R; mkdir -p $OUT
PART=${PART:-12}   # 1: rocprofv3 / PMC passes + the 256-image lines; 2: per-shape table, 512-image lines, eval forward (two calls fit 1200 s each)
if [[ $PART == *1* ]]; then
bash scripts/profile_round.sh > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?" | tee $OUT/summary.txt
cp gpurun_out/round/pmc_traffic.json profiles/${R}_pmc_traffic_b256.json
cp gpurun_out/round/kernel_stats.csv $OUT/kernel_stats.csv
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_b256.json 2> $OUT/bench_b256.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/bench_b256.err
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap > $OUT/bench_b256_no_overlap.json 2>/dev/null; echo "bench no-overlap rc=$?" | tee -a $OUT/summary.txt
MAAI_FOLD=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap > $OUT/bench_b256_no_fold_no_overlap.json 2>/dev/null; echo "bench no-fold rc=$?" | tee -a $OUT/summary.txt
fi
if [[ $PART == *2* ]]; then
timeout -k 10 400 python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --detail --profile-table $OUT/table.json > $OUT/bench_detail.json 2> $OUT/bench_detail.err; echo "bench detail rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 400 python3 bench.py --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_b512_lean.json 2> $OUT/bench_b512.err; echo "bench512 (lean, nothing recomputed) rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 400 python3 bench.py --batch 512 --recompute --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_b512_recompute_stage1.json 2> $OUT/bench_b512_s1.err; echo "bench512 stage 1 rc=$?" | tee -a $OUT/summary.txt
MAAI_RECOMPUTE_LAYERS=1,2 timeout -k 10 400 python3 bench.py --batch 512 --recompute --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_b512_recompute_stage12.json 2> $OUT/bench_b512_s12.err; echo "bench512 stages 1-2 rc=$?" | tee -a $OUT/summary.txt
MAAI_RECOMPUTE_LAYERS=1 MAAI_LEAN_ACT=0 timeout -k 10 400 python3 bench.py --batch 512 --recompute --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_b512_recompute_stage1_stored.json 2> $OUT/bench_b512_s1s.err; echo "bench512 stage 1, a1 stored rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 200 python3 scripts/eval_bench.py > $OUT/eval_bench.txt 2>&1; echo "eval bench rc=$?" | tee -a $OUT/summary.txt
fi
for f in $OUT/bench_b256.json $OUT/bench_b256_no_overlap.json $OUT/bench_b256_no_fold_no_overlap.json $OUT/bench_b512_lean.json $OUT/bench_b512_recompute_stage1.json $OUT/bench_b512_recompute_stage12.json $OUT/bench_b512_recompute_stage1_stored.json; do [ -s $f ] && python3 -c "import json,sys;d=json.load(open('$f'));print('$f',d['value'],d['ms_per_step'],d['config']['peak_hbm_GB'],d['config'].get('overlap_views'))"; done
tail -3 $OUT/eval_bench.txt
