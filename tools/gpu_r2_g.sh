#!/bin/bash
# round 2, visit G: batch-split probe, coarse dQ stamps, packed-fp32 check (compiler-generated v_pk_*)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python tools/split_probe.py > gpurun_out/g_split.log 2>&1
timeout 200 python tools/attn_trace.py > gpurun_out/g_trace.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w tools/ubench/ubench.hip -o /tmp/ubench && timeout 200 /tmp/ubench 2>&1 | tail -2 > gpurun_out/g_pkcheck.log
cat gpurun_out/g_split.log; tail -30 gpurun_out/g_trace.log; cat gpurun_out/g_pkcheck.log
