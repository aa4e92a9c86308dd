#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3final3
mkdir -p $OUT
bash scripts/profile_round.sh > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/profile_round.log | cut -c1-300
cp gpurun_out/round/pmc_traffic.json profiles/r03_pmc_traffic_b256.json
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_with_traffic.json 2> $OUT/bench_with_traffic.err; echo "bench2 rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 400 python3 bench.py --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_b512.json 2> $OUT/bench_b512.err; echo "bench512 rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 400 python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --detail --profile-table $OUT/table.json > $OUT/bench_detail.json 2> $OUT/bench_detail.err; echo "bench detail rc=$?" | tee -a $OUT/summary.txt
