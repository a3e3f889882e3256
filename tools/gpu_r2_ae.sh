#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do
  for tag in "" _lnres; do
    echo "== libpainter_hip$tag"; PAINTER_AMD_LIB=$PWD/painter_amd/lib/libpainter_hip$tag.so timeout 200 python tools/misc_bench.py 2>&1 | grep -E "ln_bwd|ln_fwd"
  done
done > gpurun_out/ae_ln.log 2>&1
PAINTER_AMD_LIB=$PWD/painter_amd/lib/libpainter_hip_lnres.so timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "layernorm or ln_" 2>&1 | tail -2 >> gpurun_out/ae_ln.log
cat gpurun_out/ae_ln.log
