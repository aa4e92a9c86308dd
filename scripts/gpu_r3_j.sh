#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3j
mkdir -p $OUT
python3 __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/smoke.log
bash scripts/pmc_conv_shape.sh 256 256 56 3 256 pp3x3_c256 fwd > $OUT/sq_pp.log 2>&1; echo "sq pp rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/sq_pp.log
MAAI_CONV_PP=0 bash scripts/pmc_conv_shape.sh 256 256 56 3 256 ring3x3_c256 fwd > $OUT/sq_ring.log 2>&1; echo "sq ring rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/sq_ring.log
bash scripts/pmc_conv_shape.sh 1024 512 56 1 256 ppw_1024x512 wgrad > $OUT/sq_ppw.log 2>&1; echo "sq ppw rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/sq_ppw.log
bash scripts/pmc_conv_shape.sh 256 256 56 3 256 wgrad3x3_c256 wgrad > $OUT/sq_wg3.log 2>&1; echo "sq wgrad3x3 rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/sq_wg3.log
bash scripts/pmc_conv_shape.sh 64 256 224 1 256 bwd3_224 bwd3 > $OUT/sq_bwd3.log 2>&1; echo "sq bwd3 rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/sq_bwd3.log
bash scripts/pmc_conv_shape.sh 64 256 224 1 256 chain_224 chain > $OUT/sq_chain.log 2>&1; echo "sq chain rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/sq_chain.log
