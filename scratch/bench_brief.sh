#!/bin/bash
# prints value / ms_per_step / roofline achieved + avg launch for N bench runs
for i in $(seq 1 ${1:-2}); do
  timeout -k 10 300 python bench.py --no-cpu-baseline ${@:2} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('achieved'), r.get('avg_launch_us'), d['config'].get('workload','')[:60])" || exit 1
done
