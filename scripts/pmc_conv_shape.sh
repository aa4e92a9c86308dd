#!/bin/bash
# SQ counters (three --pmc passes) for one convolution shape: bash scripts/pmc_conv_shape.sh CIN COUT HW K B TAG
set -e -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o runc -- python3 $REPO/scripts/conv_one.py $1 $2 $3 $4 $5 > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
dur = []
for f in glob.glob(os.path.join(out, "p*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if not any(t in r["Kernel_Name"] for t in ("conv_igemm", "conv_pws", "conv_chain")):
            continue
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {k: v[1] / v[0] for k, v in acc.items()}
res["duration_ns"] = sum(dur) / max(len(dur), 1)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
