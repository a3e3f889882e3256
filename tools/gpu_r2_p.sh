#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for pad in 0 90000; do
  echo "== PA_ATTN3_LDS_PAD=$pad" 
  PA_ATTN3_LDS_PAD=$pad timeout 200 python tools/attn_bench.py 2 2>&1 | grep "B'=8 gen3 4-wave"
  PA_ATTN3_LDS_PAD=$pad timeout 200 python tools/attn_trace.py 2>&1 | tail -5
done > gpurun_out/p_occ.log 2>&1
cat gpurun_out/p_occ.log
