#!/bin/bash
OUT=gpurun_out/r3ae
mkdir -p $OUT
for k in 0 2 4 0 3 6; do
  echo "== stagger $k" >> $OUT/ab.txt
  MAAI_PP_STAGGER=$k timeout -k 10 300 python scripts/pp_ab.py 256 c256 2>&1 | grep -v amdgpu | cut -c1-130 >> $OUT/ab.txt || exit 1
done
cat $OUT/ab.txt
