#!/bin/bash
# A/B of environment settings on a bench configuration, one process each on the same box:
#   tools/diag/ab_env.sh "<bench.py arguments>" "VAR=a VAR2=b" "VAR=c" ...   (an empty string = the defaults)
ARGS=$1; shift
run() { echo "== ${*:-defaults}"; env "$@" timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  tiles/s', d['value'], 'ms', d['ms_per_step'])"; }
for c in "$@"; do run $c; done
