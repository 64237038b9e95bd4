#!/bin/bash
# Dynamic instruction mix / stall counters of one wgrad or conv launch (diagnostic): tools/pmc_one.sh wgrad|conv
export TMPDIR=/tmp
W=${1:-wgrad}
mkdir -p gpurun_out/pmc_one
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_one/a_$W -o pmc -- python3 scratch/one_wgrad.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_one/b_$W -o pmc -- python3 scratch/one_wgrad.py $W > /dev/null 2>&1
python3 - $W <<'PY'
import csv, collections, sys
W = sys.argv[1]
for tag in ("a", "b"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_one/{tag}_{W}/pmc_counter_collection.csv")))
    acc = collections.defaultdict(list)
    for r in rows:
        if ("conv_wgrad" if W == "wgrad" else "conv_igemm") in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            name = r["Kernel_Name"][:70]; vg = r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"]
    print(name, vg)
    for k, v in sorted(acc.items()):
        print(f"   {k:28s} {sum(v)/len(v):16.0f}")
PY
