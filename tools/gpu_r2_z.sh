#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "gemm or linear or wgrad or gelu or mlp" 2>&1 | tail -4 > gpurun_out/z_tests.log
timeout 300 python tools/fc1_epilogue.py 2>&1 | grep -v amdgpu > gpurun_out/z_fc1.log
timeout 300 python tools/gemm_order.py 2>&1 | grep -E "fc2 dgrad|fc1\+gelu" >> gpurun_out/z_fc1.log
cat gpurun_out/z_tests.log gpurun_out/z_fc1.log
