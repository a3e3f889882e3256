#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/w_prof
PAINTER_AMD_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w_prof -o one -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 > gpurun_out/w_prof.log 2>&1
rm -f gpurun_out/w_prof/*kernel_trace.csv
timeout 300 python tools/attn3_diag.py 2>&1 | grep -v amdgpu > gpurun_out/w_diag.log
tail -1 gpurun_out/w_prof.log | cut -c1-200; grep -A1 "56, 28" gpurun_out/w_diag.log
