#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
