#!/bin/bash
# Build an experiment variant of the library next to the shipped one (A/B runs: MAAI_LIB_PATH=<variant> python ...).
#   scripts/build_variant.sh <name> "<extra hipcc flags, e.g. -DMAAI_EXP=1>"
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)/multimodal-active-ai_amd
OUT=$ROOT/lib/variants
mkdir -p $OUT /tmp/maai_variant_$NAME
for f in $ROOT/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function "$@" -c $f -o /tmp/maai_variant_$NAME/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmaai_hip_$NAME.so /tmp/maai_variant_$NAME/*.o
echo $OUT/libmaai_hip_$NAME.so
