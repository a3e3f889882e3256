#!/bin/bash
# round 2, visit C: micro-benchmarks (MFMA chains, exp throughput, wave pairing, barriers), in-situ kernel trace of the step with
# generation 2 vs 3, one-stream vs two-stream step, the new parity tests, the bench line
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w tools/ubench/ubench.hip -o /tmp/ubench && timeout 120 /tmp/ubench > gpurun_out/c_ubench.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/ubench/ubench.hip -o /tmp/ubench_slp && timeout 120 /tmp/ubench_slp > gpurun_out/c_ubench_slp.log 2>&1
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -k "boundary or reference_training or b8 or n32 or tight or generations" 2>&1 | tail -30 > gpurun_out/c_tests_new.log
timeout 300 python tools/step_ab.py 2 5 2,0 > gpurun_out/c_step_ab.log 2>&1
PAINTER_AMD_SIDE_STREAM=0 timeout 300 python tools/step_ab.py 2 5 2,0 > gpurun_out/c_step_ab_1stream.log 2>&1
rm -rf gpurun_out/c_prof2 gpurun_out/c_prof3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c_prof2 -o g2 -- python tools/step_ab.py 1 4 2 > gpurun_out/c_prof2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c_prof3 -o g3 -- python tools/step_ab.py 1 4 0 > gpurun_out/c_prof3.log 2>&1
rm -f gpurun_out/c_prof2/*kernel_trace.csv gpurun_out/c_prof3/*kernel_trace.csv
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/c_bench.log 2>&1
cat gpurun_out/c_ubench.log; tail -12 gpurun_out/c_tests_new.log; cat gpurun_out/c_step_ab.log gpurun_out/c_step_ab_1stream.log | grep generation; tail -1 gpurun_out/c_bench.log | cut -c1-400
