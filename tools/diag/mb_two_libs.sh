#!/bin/bash
# tools/diag/mb_two_libs.sh <other libpssr_mi355.so> [fwd|wgrad|all] [microbench script]: a per-layer microbenchmark (default
# tools/diag/microbench_conv.py) with the other build (A), then with the in-tree one (B)
set -e
A=$1; W=${2:-fwd}; MB=${3:-tools/diag/microbench_conv.py}
LIB=pssr2_amd/libpssr_mi355.so
cp $LIB /tmp/mb_lib_B.so
cp $A $LIB; echo "== A ($A)"; timeout -k 10 200 python $MB $W 2>/dev/null || true
cp /tmp/mb_lib_B.so $LIB; echo "== B (in-tree)"; timeout -k 10 200 python $MB $W 2>/dev/null
