#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_optim_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -12 > gpurun_out/tests_opt.log
cat gpurun_out/tests_opt.log
