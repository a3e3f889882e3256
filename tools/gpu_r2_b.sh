#!/bin/bash
# round 2, visit B: paired 8-wave build of the generation-3 attention: correctness breakdown, timing against the 4-wave build and
# generation 2, SQ counters, whole-step A/B, full suite, bench line
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/attn3_diag.py > gpurun_out/b_diag.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k attn -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/b_tests_attn.log
timeout 300 python tools/attn_bench.py 3 > gpurun_out/b_attn_bench.log 2>&1
rm -rf gpurun_out/b_prof gpurun_out/b_pmc1 gpurun_out/b_pmc2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b_prof -o b -- python tools/attn_bench.py 1 > gpurun_out/b_prof.log 2>&1
rm -f gpurun_out/b_prof/*kernel_trace.csv
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/b_pmc1 -o p -- python tools/attn_bench.py 1 > gpurun_out/b_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/b_pmc2 -o p -- python tools/attn_bench.py 1 > gpurun_out/b_pmc2.log 2>&1
rm -f gpurun_out/b_pmc1/*kernel_trace.csv gpurun_out/b_pmc2/*kernel_trace.csv
timeout 600 python tools/step_ab.py 3 5 2,3,0 > gpurun_out/b_step_ab.log 2>&1
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/b_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/b_bench.log 2>&1
cat gpurun_out/b_diag.log; tail -8 gpurun_out/b_tests_attn.log; cat gpurun_out/b_attn_bench.log; cat gpurun_out/b_step_ab.log; tail -5 gpurun_out/b_tests.log; tail -1 gpurun_out/b_bench.log | cut -c1-1500
