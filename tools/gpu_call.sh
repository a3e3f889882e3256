#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 > gpurun_out/tests.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1
cat gpurun_out/tests.log; tail -2 gpurun_out/smoke.log; grep metric gpurun_out/bench.log
