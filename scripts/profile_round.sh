#!/bin/bash
# One GPU-box call that produces everything profiles/ holds for a round (run from the repo root under gpurun):
#   bench line, rocprofv3 kernel-trace stats, and the two PMC traffic passes of the same command.
set -e -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
# The weight-gradient split-K budgets are measured ONCE, by an untraced run, and reused by every traced run through this
# file: the traces then hold no tuning launches (round 2's kernel_stats did: its traced run was the one that tuned).
export MAAI_WGRAD_TUNE_FILE=$OUT/wgrad_tune.json
rm -f $MAAI_WGRAD_TUNE_FILE
timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/tune_run.json 2> $OUT/tune_run.err
test -s $MAAI_WGRAD_TUNE_FILE || { echo "no tune file written"; exit 1; }
echo "tuning run done"
if [ -z "$SKIP_BENCH" ]; then timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; fi
echo "bench done"
cd /tmp && export TMPDIR=/tmp
# (the traced runs put the two forwards of a step one after the other, --no-overlap: a kernel's traced duration is then its own,
#  not its own plus whatever shared the device with it — the same setting bench.py's live per-kernel table is measured with)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o runc -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > $OUT/stats.log 2>&1
echo "stats done"
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o runc -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap > $OUT/pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o runc -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap > $OUT/pmc_write.log 2>&1
echo "write pass done"
cd $REPO
python3 scripts/profile_summary.py stats $OUT/stats $OUT/kernel_stats.csv
python3 scripts/profile_summary.py pmc $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap (warm-up + 1 step + the two instrumented steps = 4 steps; separate passes per counter)"
# the raw traces are large: keep only the summaries
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/bench.json
