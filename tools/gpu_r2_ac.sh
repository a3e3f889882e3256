#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
  echo "== default build"; timeout 200 python tools/attn_bench.py 2 2>&1 | grep -E "B'=8 (gen2|gen3 4-wave)"
  echo "== A3_PRIO=1 build"; PAINTER_AMD_LIB=$PWD/painter_amd/lib/libpainter_hip_prio.so timeout 200 python tools/attn_bench.py 2 2>&1 | grep -E "B'=8 (gen2|gen3 4-wave)"
done > gpurun_out/ac_prio.log 2>&1
cat gpurun_out/ac_prio.log
