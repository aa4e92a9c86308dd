#!/bin/bash
# SQ counters (three --pmc passes) for one kernel shape: bash scripts/pmc_conv_shape.sh CIN COUT HW K B TAG [fwd|wgrad|bwd3|chain]
set -e -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$6
MODE=${7:-fwd}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o runc -- python3 $REPO/scripts/conv_one.py $1 $2 $3 $4 $5 $MODE > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
cd $REPO
python3 - "$OUT" "$*" <<'PY'
import csv, glob, os, sys, json, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
dur = []
names = collections.Counter()
for f in glob.glob(os.path.join(out, "p*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if not any(t in r["Kernel_Name"] for t in ("conv_igemm", "conv_pws", "conv_chain", "conv_pp", "conv_c64", "conv_bwd3", "wgrad")):
            continue
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            names[r["Kernel_Name"].split("(")[0][:90]] += 1
res = {k: v[1] / v[0] for k, v in acc.items()}
d = sum(dur) / max(len(dur), 1)
wc = res.get("SQ_WAVE_CYCLES", 0.0) or 1.0
doc = {"command": "rocprofv3 --pmc <set> --kernel-trace -- python3 scripts/conv_one.py " + sys.argv[2] + "  (three passes, SQ counters only; scripts/pmc_conv_shape.sh)",
       "kernels": dict(names), "mean_per_launch": dict(res, duration_ns=d),
       "derived": {"mfma_pipe_utilisation_at_2.1GHz": res.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d * 2.1 * 1024) if d else None,
                   "lds_bank_conflict_fraction": res.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(res.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0),
                   "wave_cycle_split": {"issue_stall(WAIT_INST_ANY)": res.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                                        "waitcnt_or_barrier(WAIT_ANY)": res.get("SQ_WAIT_ANY", 0.0) / wc,
                                        "issuing(ACTIVE_INST_ANY)": res.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}}}
json.dump(doc, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(doc["derived"], indent=1), d)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
