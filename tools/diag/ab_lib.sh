#!/bin/bash
# A/B of two builds of the library on the same box: tools/diag/ab_lib.sh <other libpssr_mi355.so> [rounds] [extra bench.py arguments]
# (the in-tree library is B; each round runs the training bench with A, then with B)
set -e
A=$1; R=${2:-2}; shift; shift || true
LIB=pssr2_amd/libpssr_mi355.so
cp $LIB /tmp/ab_lib_B.so
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 30 --warmup 5 --tiles 1280 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $tag tiles/s', d['value'], 'ms', d['ms_per_step'], 'conv TF', d['roofline']['achieved'])"; }
for i in $(seq $R); do
    cp $A $LIB; run A "$@"
    cp /tmp/ab_lib_B.so $LIB; run B "$@"
done
