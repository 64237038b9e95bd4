#!/bin/bash
# A/B of two TREES on the same box: tools/diag/ab_tree.sh <other tree (holding bench.py + pssr2_amd/ with its built library)> <rounds> [bench.py arguments]
# Each round runs the bench in the other tree (A), then in this one (B).
A=$1; R=${2:-2}; shift; shift
HERE=$(pwd)
run() { tag=$1; dir=$2; shift; shift; (cd $dir && timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $tag tiles/s', d['value'], 'ms', d['ms_per_step'])"); }
for i in $(seq $R); do
    run A $A "$@" || exit 1
    run B $HERE "$@" || exit 1
done
