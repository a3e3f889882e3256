#!/bin/bash
# round 2, visit A: generation-3 attention -- correctness breakdown, A/B timing against generation 2, kernel trace + SQ counters,
# then the full GPU suite and the bench line
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/attn3_diag.py > gpurun_out/a_diag.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k attn -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/a_tests_attn.log
timeout 300 python tools/attn_bench.py 3 > gpurun_out/a_attn_bench.log 2>&1
rm -rf gpurun_out/a_prof gpurun_out/a_pmc1 gpurun_out/a_pmc2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/a_prof -o a -- python tools/attn_bench.py 1 > gpurun_out/a_prof.log 2>&1
rm -f gpurun_out/a_prof/*kernel_trace.csv
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/a_pmc1 -o p -- python tools/attn_bench.py 1 > gpurun_out/a_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/a_pmc2 -o p -- python tools/attn_bench.py 1 > gpurun_out/a_pmc2.log 2>&1
rm -f gpurun_out/a_pmc1/*kernel_trace.csv gpurun_out/a_pmc2/*kernel_trace.csv
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/a_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/a_bench.log 2>&1
PA_ATTN3=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench_gen2.log 2>&1
cat gpurun_out/a_diag.log; tail -8 gpurun_out/a_tests_attn.log; cat gpurun_out/a_attn_bench.log; tail -5 gpurun_out/a_tests.log; tail -2 gpurun_out/a_bench.log | cut -c1-600; tail -1 gpurun_out/a_bench_gen2.log | cut -c1-300
