#!/bin/bash
OUT=gpurun_out/r3ah
mkdir -p $OUT
V=$PWD/multimodal-active-ai_amd/lib/variants
for v in base colmajor base colmajor; do
  echo "== $v" >> $OUT/ab.txt
  if [ $v = base ]; then timeout -k 10 200 python scripts/pp_ab.py 256 c64 2>&1 | grep -v amdgpu | cut -c1-130 >> $OUT/ab.txt || exit 1
  else MAAI_LIB_PATH=$V/libmaai_hip_$v.so timeout -k 10 200 python scripts/pp_ab.py 256 c64 2>&1 | grep -v amdgpu | cut -c1-130 >> $OUT/ab.txt || exit 1; fi
done
cat $OUT/ab.txt
