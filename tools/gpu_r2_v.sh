#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attn" 2>&1 | tail -3 > gpurun_out/v_tests.log
timeout 300 python tools/attn_bench.py 2 2>&1 | grep "gen3 4-wave" > gpurun_out/v_attn.log
timeout 300 python tools/attn_ablate.py epilogue 2>&1 | tail -9 >> gpurun_out/v_attn.log
cat gpurun_out/v_tests.log gpurun_out/v_attn.log
