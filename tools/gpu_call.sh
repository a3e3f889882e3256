#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for e in 1 2 1 2; do PAINTER_AMD_SIDE_STREAMS=$e timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side streams $e', d['value'], d['ms_per_step'])"; done
PAINTER_AMD_SIDE_STREAMS=2 PA_WGRAD_WGS=32 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 streams, 32 WGs', d['value'], d['ms_per_step'])"
