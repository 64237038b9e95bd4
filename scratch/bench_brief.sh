#!/bin/bash
# prints value / ms_per_step / roofline achieved + avg launch for N bench runs
for i in $(seq 1 ${1:-2}); do
  timeout -k 10 200 python bench.py --no-cpu-baseline ${@:2} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_us'])" || exit 1
done
