#!/bin/bash
# round 2, visit S: full GPU suite, smoke, bench line, kernel trace + PMC passes of the bench, packed-fp32 check
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/s_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 > gpurun_out/s_smoke.log
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/s_bench.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w tools/ubench/ubench.hip -o /tmp/ubench && timeout 200 /tmp/ubench 2>&1 | tail -3 > gpurun_out/s_pkcheck.log
rm -rf gpurun_out/s_prof gpurun_out/s_pmc_f gpurun_out/s_pmc_w gpurun_out/s_pmc_m
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s_prof -o r2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 > gpurun_out/s_prof.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/s_pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 > gpurun_out/s_pmc_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/s_pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 > gpurun_out/s_pmc_w.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/s_pmc_m -o m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 > gpurun_out/s_pmc_m.log 2>&1
python tools/pmc_traffic.py gpurun_out/s_pmc_f gpurun_out/s_pmc_w gpurun_out/s_roofline_traffic.json > gpurun_out/s_traffic.log 2>&1
python tools/pmc_mfma.py gpurun_out/s_pmc_m gpurun_out/s_pmc_mfma_busy_per_kernel.csv > gpurun_out/s_mfma.log 2>&1
rm -f gpurun_out/s_prof/*kernel_trace.csv gpurun_out/s_pmc_f/*kernel_trace.csv gpurun_out/s_pmc_w/*kernel_trace.csv gpurun_out/s_pmc_m/*kernel_trace.csv gpurun_out/s_pmc_*/*counter_collection.csv
tail -6 gpurun_out/s_tests.log; cat gpurun_out/s_smoke.log gpurun_out/s_pkcheck.log; tail -1 gpurun_out/s_bench.log | cut -c1-300; cat gpurun_out/s_mfma.log
