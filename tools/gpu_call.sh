#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/ddp_selftest.py > gpurun_out/ddp.log 2>&1
tail -4 gpurun_out/ddp.log
PAINTER_AMD_DDP_SELFTEST=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_ddp1.log 2>&1
tail -1 gpurun_out/bench_ddp1.log | cut -c1-200
