#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/attn_trace.py 8 > gpurun_out/d_trace.log 2>&1
for w in 64 96 128 192 256; do PA_WGRAD_WGS=$w timeout 200 python tools/step_ab.py 2 5 0 2>&1 | grep generation | sed "s/^/wgrad_wgs=$w /" >> gpurun_out/d_wgs.log; done
cat gpurun_out/d_trace.log gpurun_out/d_wgs.log
