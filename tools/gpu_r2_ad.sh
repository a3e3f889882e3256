#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
  for tag in "" _nohoist _nondl; do
    echo "== build libpainter_hip$tag"; PAINTER_AMD_LIB=$PWD/painter_amd/lib/libpainter_hip$tag.so timeout 200 python tools/attn_bench.py 2 2>&1 | grep -E "B'=8 gen3 4-wave"
  done
done > gpurun_out/ad_variants.log 2>&1
cat gpurun_out/ad_variants.log
