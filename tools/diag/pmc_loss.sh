#!/bin/bash
# instruction mix / stall counters of the MS-SSIM kernels (two PMC passes over tools/diag/microbench_loss.py)
export TMPDIR=/tmp
O=gpurun_out/pmc_loss; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -o pmc -- python3 tools/diag/microbench_loss.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/b -o pmc -- python3 tools/diag/microbench_loss.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for sub in "ab":
    f = glob.glob(f"gpurun_out/pmc_loss/{sub}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ssim" not in k: continue
        key = k.split("(")[0][-22:] + " grid" + r["Grid_Size_X"] if "Grid_Size_X" in r else k[:40]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
    for key in sorted(acc):
        print(key, {c: round(v / n[(key, c)]) for c, v in acc[key].items()})
PY
