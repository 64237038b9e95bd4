#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the convolution microbenchmark's launches (two rocprofv3 PMC passes), per (kernel, grid):
#   tools/diag/pmc_fetch_mb.sh <out dir> [microbench argument: fwd|wgrad|all]
# environment (PSSR_* tunables) is inherited, so two settings can be compared
set -e
O=$1; W=${2:-fwd}
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o pmc -- python3 tools/diag/microbench_conv.py $W > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o pmc -- python3 tools/diag/microbench_conv.py $W > $O/w.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
def load(sub, ctr):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{o}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            k = (r["Kernel_Name"][:90], r["Grid_Size"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
fe, wr = load("f", "FETCH_SIZE"), load("w", "WRITE_SIZE")
print("kernel | grid | launches | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | MB/launch = (2 x fetch + write)")
for k in sorted(fe, key=lambda k: -fe[k][0]):
    f = fe[k][0] / fe[k][1]; w = wr[k][0] / wr[k][1] if k in wr and wr[k][1] else 0.0
    if f + w < 2000: continue
    print(f"{k[0]} | {k[1]} | {fe[k][1]} | {f:10.0f} | {w:10.0f} | {(2*f+w)*1024/1e6:8.1f}")
PY
