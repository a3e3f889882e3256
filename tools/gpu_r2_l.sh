#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pair_pipeline_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/l_tests.log
timeout 300 python tools/pair_pipeline_bench.py > gpurun_out/l_bench.log 2>&1
cat gpurun_out/l_tests.log; tail -2 gpurun_out/l_bench.log
