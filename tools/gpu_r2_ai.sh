#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for v in 16 4; do
  echo -n "PA_RELPOS_SPLITS=$v  "; PA_RELPOS_SPLITS=$v timeout 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-optimizer --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
done > gpurun_out/ai_relpos.log 2>&1
cat gpurun_out/ai_relpos.log
echo -n "default (engine asks for 4)  "; timeout 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-optimizer --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x -k "attn_bwd or small_fp32 or small_bf16" 2>&1 | tail -2
