#!/bin/bash
# round 2, closing check: whole GPU suite, smoke, bench line
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/final_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 > gpurun_out/final_smoke.log
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench.log 2>&1
grep -E "passed|failed" gpurun_out/final_tests.log; cat gpurun_out/final_smoke.log; tail -1 gpurun_out/final_bench.log | cut -c1-260
