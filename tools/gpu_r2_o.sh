#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "gemm or linear or wgrad" 2>&1 | tail -4 > gpurun_out/o_tests.log
timeout 300 python tools/gemm_order.py > gpurun_out/o_order.log 2>&1
timeout 300 python tools/attn3_diag.py > gpurun_out/o_diag.log 2>&1
cat gpurun_out/o_tests.log gpurun_out/o_order.log | grep -v amdgpu.ids; tail -40 gpurun_out/o_diag.log
