#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for e in 0 1 0 1; do PAINTER_AMD_SIDE_EXTRA=$e timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side extra $e', d['value'], d['ms_per_step'])"; done
PAINTER_AMD_DDP_SELFTEST=1 timeout 300 python tools/ddp_selftest.py 2>&1 | grep "^world" | cut -c1-250
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
