#!/bin/bash
# Instruction mix / MFMA-busy / LDS-conflict counters of the convolution and weight-gradient kernels (two PMC passes over
# tools/diag/microbench_conv.py at the c2 layer shapes, batch 32).  Prints one line per (kernel, grid); tee it into profiles/.
export TMPDIR=/tmp
O=gpurun_out/pmc_conv; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/a -o pmc -- python3 tools/diag/microbench_conv.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/b -o pmc -- python3 tools/diag/microbench_conv.py > $O/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
for sub in "ab":
    f = glob.glob(f"gpurun_out/pmc_conv/{sub}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"\d\d(conv_(?:v3|igemm|flat|wgrad16d|wgrad16|wgrad16_1x1)_kernel)I(.*?)EEvN", k)
        if not m: continue
        key = m.group(1) + " " + m.group(2).replace("DF16b", "bf16 ").replace("Li", "").replace("E", " ").strip() + " | grid " + r["Grid_Size"]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel / grid':66s} {'us':>7s} {'VALU/MFMA':>9s} {'MFMA busy':>9s} {'LDS confl':>9s} {'wait':>6s}   (VALU/MFMA: other vector instructions per MFMA;")
print(f"{'':66s} {'':>7s} {'':>9s} {'':>9s} {'':>9s} {'':>6s}    MFMA busy: 32 clk x MFMAs / (1024 SIMDs x GRBM_GUI_ACTIVE/8); LDS confl: conflict / LDS-array cycles)")
for key in sorted(acc):
    c = {k: v / n[(key, k)] for k, v in acc[key].items()}
    if not c.get("SQ_INSTS_MFMA"): continue
    clk = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{key[:66]:66s} {sum(dur[key]) / len(dur[key]):7.1f} {c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA'] - 1:9.2f} "
          f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk):9.3f} {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):9.3f} "
          f"{c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):6.3f}")
    print("      raw:", {k: round(v) for k, v in sorted(c.items())})
PY
