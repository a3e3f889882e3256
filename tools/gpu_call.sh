#!/bin/bash
# scratch GPU visit (edited per call); outputs in gpurun_out/
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "conv64" 2>&1 | tail -15 > gpurun_out/tests_conv.log
timeout 300 python tools/conv_bench.py > gpurun_out/conv_bench.log 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 > gpurun_out/tests_model.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
cat gpurun_out/tests_conv.log gpurun_out/conv_bench.log gpurun_out/tests_model.log; tail -2 gpurun_out/bench.log
