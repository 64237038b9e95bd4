#!/bin/bash
# Profiles bench.py on the MI355X box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (hipGraph replay) and of the eager launch mode
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) in eager mode, few steps
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profile.py turns them into profiles/<round>_*.
set -e
TAG=${1:-r01}
MODE=${2:-train}
MODEL=${3:-resunet}
EXTRA=""
KEY=$MODE
if [ "$MODEL" != "resunet" ]; then EXTRA="--model $MODEL --crappifier poisson"; KEY=${MODEL}_${MODE}; fi
OUT=gpurun_out/prof_${TAG}_${KEY}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -o trace -- python3 bench.py --mode $MODE --steps 10 --warmup 3 --tiles 1024 --no-extras --no-cpu-baseline --tile-workers 1 $EXTRA > $OUT/bench_graph.json 2> $OUT/bench_graph.err
echo "graph trace done"
# eager trace and PMC passes: one kernel at a time (PSSR_WGRAD_STREAM=0 keeps the weight-gradient kernels on the launch stream), so that
# per-kernel durations and counters belong to that kernel alone; the graph trace above is the real, overlapped step
export PSSR_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o trace -- python3 bench.py --mode $MODE --steps 5 --warmup 2 --tiles 512 --no-graph --no-extras --no-cpu-baseline --tile-workers 1 $EXTRA > $OUT/bench_eager.json 2> $OUT/bench_eager.err
echo "eager trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --mode $MODE --steps 2 --warmup 1 --tiles 256 --no-graph --no-extras --no-cpu-baseline --tile-workers 1 $EXTRA > $OUT/bench_pmc_fetch.json 2> $OUT/bench_pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --mode $MODE --steps 2 --warmup 1 --tiles 256 --no-graph --no-extras --no-cpu-baseline --tile-workers 1 $EXTRA > $OUT/bench_pmc_write.json 2> $OUT/bench_pmc_write.err
echo "pmc write done"
# keep the merge-back small: per-dispatch traces can be large
find $OUT -name '*.csv' -size +20M -delete
du -sh $OUT
