#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attn" 2>&1 | tail -5 > gpurun_out/k_tests.log
timeout 300 python tools/attn_bench.py 2 > gpurun_out/k_attn.log 2>&1
timeout 200 python tools/attn_trace.py 2>&1 | tail -4 > gpurun_out/k_trace.log
cat gpurun_out/k_tests.log gpurun_out/k_attn.log gpurun_out/k_trace.log | grep -v amdgpu.ids
