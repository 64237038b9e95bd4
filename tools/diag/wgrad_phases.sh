#!/bin/bash
# timing experiments on the 3x3 weight-gradient kernel: workgroup count and skipped phases
for blocks in 256 512 1024; do
for dbg in 0 16 32 48; do
echo "== WGRAD_BLOCKS=$blocks IGEMM_DBG=$dbg"
PSSR_WGRAD_BLOCKS=$blocks PSSR_IGEMM_DBG=$dbg timeout -k 10 100 python tools/diag/microbench_conv.py wgrad 2>&1 | grep -E "^L0|^L2|^head" | cut -c1-20,65-130
done; done
