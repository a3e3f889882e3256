#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w tools/ubench/ubench.hip -o /tmp/ubench && timeout 200 /tmp/ubench > gpurun_out/h_ubench.log 2>&1
tail -16 gpurun_out/h_ubench.log
