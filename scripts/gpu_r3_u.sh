#!/bin/bash
OUT=gpurun_out/r3u
mkdir -p $OUT
for r in 3 11 3 11; do
echo "RULE=$r" >> $OUT/eval.txt
MAAI_CONV_PP_RULE=$r timeout -k 10 300 python3 scripts/eval_bench.py 2>&1 | grep "65536 rows + fast" >> $OUT/eval.txt || exit 1
done
cat $OUT/eval.txt
