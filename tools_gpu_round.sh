#!/bin/bash
# one GPU visit: tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/tests.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 ${BENCH_FLAGS:-} > gpurun_out/bench.log 2>&1
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1
ls -R gpurun_out/prof | head -30
tail -5 gpurun_out/tests.log; cat gpurun_out/smoke.log | tail -3; tail -3 gpurun_out/bench.log
