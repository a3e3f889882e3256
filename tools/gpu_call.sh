#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "streamk or gemm256 or linear" 2>&1 | tail -15 > gpurun_out/tests_sk.log
cat gpurun_out/tests_sk.log
timeout 300 python tools/gemm_probe.py sk > gpurun_out/probe_sk.log 2>&1
cat gpurun_out/probe_sk.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-400
