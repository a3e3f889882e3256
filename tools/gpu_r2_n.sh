#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_optim_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 > gpurun_out/n_tests.log
rm -rf gpurun_out/n_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/n_prof -o n2 -- python tools/pair_pipeline_bench.py --iters 20 --cpu-iters 1 > gpurun_out/n_prof.log 2>&1
rm -f gpurun_out/n_prof/*kernel_trace.csv
cat gpurun_out/n_tests.log; tail -1 gpurun_out/n_prof.log; ls gpurun_out/n_prof
