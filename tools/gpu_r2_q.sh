#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "2 2" "3 2" "2 3" "3 3"; do
  set -- $cfg
  echo "== DQ_WAVES=$1 DKV_WAVES=$2"
  PA_ATTN3_DQ_WAVES=$1 PA_ATTN3_DKV_WAVES=$2 timeout 200 python tools/attn_bench.py 2 2>&1 | grep "gen3 4-wave"
done > gpurun_out/q_waves.log 2>&1
PA_ATTN3_DQ_WAVES=3 PA_ATTN3_DKV_WAVES=3 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attn" 2>&1 | tail -3 >> gpurun_out/q_waves.log
cat gpurun_out/q_waves.log
