#!/bin/bash
# A/B of tunables on the c2 training bench (one process each, same box)
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --tiles 1280 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  tiles/s', d['value'], 'ms', d['ms_per_step'], 'conv TF', d['roofline']['achieved'])"; }
for c in "$@"; do run $c; done
