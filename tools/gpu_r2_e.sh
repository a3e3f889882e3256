#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 200 python tools/attn3_diag.py 2>&1 | grep -v amdgpu > gpurun_out/e_diag.log
timeout 200 python tools/attn_bench.py 3 2>&1 | grep -v amdgpu > gpurun_out/e_attn_bench.log
timeout 300 python tools/step_ab.py 3 5 0,5 2>&1 | grep generation > gpurun_out/e_step_ab.log
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -k attn -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/e_tests_attn.log
cat gpurun_out/e_diag.log gpurun_out/e_attn_bench.log gpurun_out/e_step_ab.log gpurun_out/e_tests_attn.log
