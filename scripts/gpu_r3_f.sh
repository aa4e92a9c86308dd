#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3f
mkdir -p $OUT
V=$PWD/multimodal-active-ai_amd/lib/variants
for v in base exp8 exp16 exp32; do
  if [ $v = base ]; then unset MAAI_LIB_PATH; else export MAAI_LIB_PATH=$V/libmaai_hip_$v.so; fi
  timeout -k 10 200 python scripts/pp_ab.py 256 c256 2>&1 | grep -v amdgpu | head -4 > $OUT/pp_$v.txt; echo "$v rc=$?" | tee -a $OUT/summary.txt
  sed "s/^/$v: /" $OUT/pp_$v.txt | cut -c1-140
done
