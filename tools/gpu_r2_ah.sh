#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attn" 2>&1 | tail -2 > gpurun_out/ah.log
timeout 300 python tools/attn_ablate.py epilogue 2>&1 | grep -E "^abl" | tail -9 >> gpurun_out/ah.log
cat gpurun_out/ah.log
