#!/bin/bash
# PMC comparison of the two conv main loops on one layer (diagnostic)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_one
for m in 0 2; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_one/m$m -o pmc -- python3 scratch/one_conv.py $m > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections
for m in (0, 2):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_one/m{m}/pmc_counter_collection.csv")))
    acc = collections.defaultdict(list)
    for r in rows:
        if "conv_igemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            acc["dur_us"].append(dur)
            name = r["Kernel_Name"][:60]; vg = r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"]
    print("mode", m, name, vg)
    for k, v in sorted(acc.items()):
        print(f"   {k:24s} {sum(v)/len(v):16.0f}")
PY
