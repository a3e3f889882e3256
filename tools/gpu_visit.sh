#!/bin/bash
# One parameterised command list for a GPU visit (replaces round 2's per-visit scripts).  Usage on the box, from the repo root:
#   bash tools/gpu_visit.sh <section> [<section> ...]      sections: tests newtests bench huge profile pmc overlap ab ... r5tests yardstick lightab prof1 prof2
# Everything lands under gpurun_out/<section>*; copy what should be judged into profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0                     # a faulting GPU makes every python process abort: without this each abort writes a multi-GB core (minutes)
# a faulty box shows up in the first H2D copy: check once, and do nothing else on such a box (round 3 lost 50 GPU-minutes to one)
if ! timeout 120 python -c "import torch; x = torch.randn(1 << 20).cuda(); assert abs(float((x * 2).sum()) - 2 * float(x.sum())) < 1e-2; torch.cuda.synchronize(); print('gpu sanity ok', torch.cuda.get_device_name(0))"; then
  echo "GPU SANITY CHECK FAILED -- not running anything on this box"; exit 3
fi
HEAD=${PAINTER_AMD_GIT_HEAD:-unknown}
export PAINTER_AMD_GIT_HEAD=$HEAD
# which library this visit measures: tools/adopt_profiles.py refuses to file anything from gpurun_out/ under profiles/ unless this hash is
# the hash of the library the tree builds at that moment (round 6: the closing artefacts of rounds 4 and 5 trailed the closing commit)
echo "$(sha256sum painter_amd/lib/libpainter_hip.so | cut -c1-16) $HEAD" > gpurun_out/visit_lib_sha16.txt
for s in "$@"; do
  case $s in
    tests)     timeout 480 python -m pytest tests -m gpu -q -s > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/tests.log ;;
    newtests)  timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -s -k "head_dim_80 or h14 or loss_variants or patch_embed or vit_large or c_abi" > gpurun_out/newtests.log 2>&1; echo "newtests rc=$?"; tail -15 gpurun_out/newtests.log ;;
    bench)     timeout 330 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench.json ;;
    benchfast) timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-optimizer > gpurun_out/benchfast.json 2> gpurun_out/benchfast.err; echo "benchfast rc=$?"; cut -c1-400 gpurun_out/benchfast.json ;;
    huge)      timeout 200 python bench.py --model vit_huge --steps 5 --warmup 2 --no-cpu-baseline --min-seconds 2 > gpurun_out/bench_huge.json 2> gpurun_out/bench_huge.err; echo "huge rc=$?"; cut -c1-600 gpurun_out/bench_huge.json; tail -3 gpurun_out/bench_huge.err ;;
    profile)   (cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_two_stream -o two -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-optimizer --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/prof_two.log 2>&1
                PAINTER_AMD_SIDE_STREAM=0 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_one_stream -o one -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-optimizer --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/prof_one.log 2>&1); echo "profile done"; ls gpurun_out/prof_one_stream gpurun_out/prof_two_stream 2>/dev/null | head ;;
    pmc)       (cd /tmp && for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
                  n=$(echo $c | cut -d' ' -f1); PAINTER_AMD_SIDE_STREAM=0 timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OLDPWD/gpurun_out/pmc_$n -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/pmc_$n.log 2>&1; done); echo "pmc done" ;;
    overlap)   PAINTER_AMD_DDP_SELFTEST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 600 python tools/gradsync_overlap.py > gpurun_out/overlap.log 2>&1; echo "overlap rc=$?"; tail -12 gpurun_out/overlap.log ;;
    ilv)       PAINTER_AMD_LIB=painter_amd/lib/libpainter_hip_ilv.so timeout 900 python tools/gemm_ilv_ab.py step > gpurun_out/ilv.log 2>&1; echo "ilv rc=$?"; tail -40 gpurun_out/ilv.log ;;
    power)     PAINTER_AMD_LIB=painter_amd/lib/libpainter_hip_ilv.so timeout 600 python tools/power_probe.py > gpurun_out/power.log 2>&1; echo "power rc=$?"; tail -12 gpurun_out/power.log ;;
    ilvprof)   (cd /tmp && for i in 0 2; do PA_G256_ILV=$i PAINTER_AMD_LIB=$OLDPWD/painter_amd/lib/libpainter_hip_ilv.so PAINTER_AMD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_ilv$i -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-optimizer --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/prof_ilv$i.log 2>&1; done); echo "ilvprof done"; ls gpurun_out/prof_ilv0 gpurun_out/prof_ilv2 | head ;;
    fixtests)  timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "patch_embed or small or h14_fp32 or seggpt" > gpurun_out/fixtests.log 2>&1; echo "fixtests rc=$?"; tail -5 gpurun_out/fixtests.log ;;
    knobs)     timeout 600 python tools/knob_sweep.py > gpurun_out/knobs.log 2>&1; echo "knobs rc=$?"; tail -9 gpurun_out/knobs.log ;;
    seggpt)    timeout 600 python tools/seggpt_bench.py > gpurun_out/seggpt.log 2>&1; echo "seggpt rc=$?"; tail -4 gpurun_out/seggpt.log ;;
    pmcattn)   (cd /tmp && for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
                  n=$(echo $c | cut -d' ' -f1); timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OLDPWD/gpurun_out/pmcattn_$n -o pmc -- env PYTHONPATH=$OLDPWD python $OLDPWD/tools/attn_bench.py > $OLDPWD/gpurun_out/pmcattn_$n.log 2>&1; done); echo "pmcattn done" ;;
    finaltests) timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -s -x -k "gemm256 or vit_large_b8 or abs_pos" > gpurun_out/finaltests.log 2>&1; echo "finaltests rc=$?"; tail -4 gpurun_out/finaltests.log ;;
    lnpatch)   timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_parallel_gpu.py tests/test_boundary_gpu.py -m gpu -q -s -x -k "layernorm or small or h14_fp32 or b8_train_bf16 or (two_ranks and GradSync) or bare_module" > gpurun_out/lnpatch.log 2>&1; echo "lnpatch rc=$?"; tail -4 gpurun_out/lnpatch.log ;;
    r4tests)   timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_reference_engine_gpu.py tests/test_parallel_gpu.py -m gpu -q -s -x -k "attn or gemm256 or layernorm or small or b8_train_bf16 or vit_large_b1 or reference or bench_multi or c_abi" > gpurun_out/r4tests.log 2>&1; echo "r4tests rc=$?"; tail -8 gpurun_out/r4tests.log ;;
    r4ab)      timeout 600 python tools/r04_ab.py 3 6 > gpurun_out/r4ab.log 2>&1; echo "r4ab rc=$?"; tail -8 gpurun_out/r4ab.log ;;
    attnbench) timeout 300 python tools/attn_bench.py > gpurun_out/attnbench.log 2>&1; echo "attnbench rc=$?"; tail -12 gpurun_out/attnbench.log ;;
    a4tests)   timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -x -k "attn" > gpurun_out/a4tests.log 2>&1; echo "a4tests rc=$?"; grep -a "generation\|passed\|failed\|Error\|error" gpurun_out/a4tests.log | tail -30 ;;
    r5tests)   timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r5tests.log 2>&1; echo "r5tests rc=$?"; grep -a "passed\|failed\|rror" gpurun_out/r5tests.log | tail -8 ;;
    yardstick) timeout 600 python tools/grad_yardstick.py gpurun_out/grad_yardstick.json > gpurun_out/grad_yardstick.log 2>&1; echo "yardstick rc=$?"; tail -5 gpurun_out/grad_yardstick.log ;;
    lightab)   timeout 400 python tools/step_knob_ab.py 5 6 "light last on:8=2" "light last off:8=1" > gpurun_out/lightab.log 2>&1; echo "lightab rc=$?"; tail -3 gpurun_out/lightab.log ;;
    prof1)     (cd /tmp && PAINTER_AMD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_one_stream -o one -- python $OLDPWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/prof_one.log 2>&1); echo "prof1 done"; ls gpurun_out/prof_one_stream | head -3 ;;
    prof2)     (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_two_stream -o two -- python $OLDPWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/prof_two.log 2>&1); echo "prof2 done" ;;
    wgsweep)   timeout 500 python tools/step_knob_ab.py 4 5 "wgrad target 48:3=48" "wgrad target 64:3=64" "wgrad target 80:3=80" "wgrad target 96:3=96" "wgrad target 112:3=112" > gpurun_out/wgsweep.log 2>&1; echo "wgsweep rc=$?"; tail -5 gpurun_out/wgsweep.log ;;
    pmc5)      (cd /tmp && for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
                  n=$(echo $c | cut -d' ' -f1); PAINTER_AMD_SIDE_STREAM=0 timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OLDPWD/gpurun_out/pmc_$n -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/pmc_$n.log 2>&1; done)
               python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/roofline_traffic.json > gpurun_out/pmc_traffic.log 2>&1; echo "traffic rc=$?"
               python tools/pmc_mfma.py gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/pmc_mfma_busy_per_kernel.csv > gpurun_out/pmc_mfma.log 2>&1; echo "mfma rc=$?"; head -14 gpurun_out/pmc_mfma.log
               rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES ;;
    splitab)   timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm256 or wgrad or linear" > gpurun_out/splitab_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/splitab_tests.log
               timeout 500 python tools/step_knob_ab.py 5 6 "one XCD run per K split (round 5):2=0" "split = blockIdx.y (before):2=2" > gpurun_out/splitab.log 2>&1; echo "splitab rc=$?"; tail -3 gpurun_out/splitab.log ;;
    wgsweep2)  timeout 500 python tools/step_knob_ab.py 4 5 "wgrad target 64:3=64" "wgrad target 96:3=96" "wgrad target 128:3=128" "wgrad target 192:3=192" "wgrad target 256:3=256" > gpurun_out/wgsweep2.log 2>&1; echo "wgsweep2 rc=$?"; tail -5 gpurun_out/wgsweep2.log ;;
    smoke)     timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log ;;
    hostq)     timeout 300 python tools/host_enqueue.py > gpurun_out/hostq.log 2>&1; echo "hostq rc=$?"; tail -6 gpurun_out/hostq.log ;;
    lnrows)    timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "layernorm" > gpurun_out/lnrows_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/lnrows_tests.log
               timeout 300 python tools/ln_bwd_bench.py > gpurun_out/lnrows_bench.log 2>&1; echo "bench rc=$?"; cat gpurun_out/lnrows_bench.log
               timeout 500 python tools/step_knob_ab.py 5 6 "LN backward rows split over waves (round 5):10=0" "one wave per row (before):10=1" > gpurun_out/lnrows_ab.log 2>&1; echo "ab rc=$?"; tail -3 gpurun_out/lnrows_ab.log ;;
    lnvar)     timeout 600 python tools/step_knob_ab.py 5 6 "LN bwd 4-row batches, 3 per CU:10=0" "one wave per row:10=1" "2-row batches, 4 per CU:10=2" "2-row batches, 5 per CU:10=3" > gpurun_out/lnvar_ab.log 2>&1; echo "ab rc=$?"; tail -5 gpurun_out/lnvar_ab.log ;;
    lastcheck) timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_parallel_gpu.py -m gpu -q -k "layernorm or bench or GradSync" > gpurun_out/lastcheck.log 2>&1; echo "lastcheck rc=$?"; tail -3 gpurun_out/lastcheck.log ;;
    selftest)  PAINTER_AMD_DDP_SELFTEST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python tools/ddp_selftest.py > gpurun_out/selftest.log 2>&1; echo "selftest rc=$?"; tail -4 gpurun_out/selftest.log ;;
    profhuge)  (cd /tmp && PAINTER_AMD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_huge -o huge -- python $OLDPWD/bench.py --model vit_huge --steps 3 --warmup 1 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/prof_huge.log 2>&1); echo "profhuge done"; ls gpurun_out/prof_huge | head -3 ;;
    hd80)      timeout 100 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "head_dim_80 or h14" > gpurun_out/hd80.log 2>&1; rc=$?; echo "hd80 rc=$rc"; tail -2 gpurun_out/hd80.log; [ $rc -eq 0 ] || exit 1 ;;
    pmctraffic) (cd /tmp && for c in "FETCH_SIZE" "WRITE_SIZE"; do
                  PAINTER_AMD_SIDE_STREAM=0 timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OLDPWD/gpurun_out/pmc_$c -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/pmc_$c.log 2>&1; done)
               python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/roofline_traffic.json > gpurun_out/pmc_traffic.log 2>&1; echo "traffic rc=$?"
               rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE ;;
    adopt)     cp gpurun_out/roofline_traffic.json profiles/roofline_traffic.json && echo "adopted the PMC traffic of this library for the bench line of this visit" ;;
    engab)     timeout 600 python tools/step_engine_ab.py 5 6 "delta in dQ + colsum in epilogue (default):_ATTN_PREP=fused,_FC1_COLSUM=epilogue" "prep launch:_ATTN_PREP=launch,_FC1_COLSUM=epilogue" "separate fc1 column sums:_ATTN_PREP=fused,_FC1_COLSUM=separate" > gpurun_out/engab.log 2>&1; echo "engab rc=$?"; tail -4 gpurun_out/engab.log ;;
    libab)     timeout 900 python tools/step_lib_ab.py 3 6 "round-5 build=painter_amd/lib/libpainter_hip.so" "baseline build=painter_amd/lib/libpainter_hip_base.so" > gpurun_out/libab.log 2>&1; echo "libab rc=$?"; tail -4 gpurun_out/libab.log ;;
    r5quick)   timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -s -x -k "conv64 or decoder_tail or small or vitl_b8 or vit_large_b8 or gemm256" > gpurun_out/r5quick.log 2>&1; echo "r5quick rc=$?"; grep -a "passed\|failed\|rror" gpurun_out/r5quick.log | tail -5 ;;
    r6attn)    timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -s -x -k "attn or vit_large or small_painter or small_seggpt" > gpurun_out/r6attn.log 2>&1; echo "r6attn rc=$?"; grep -a "passed\|failed\|rror\|worst" gpurun_out/r6attn.log | tail -12 ;;
    libab6)    timeout 900 python tools/step_lib_ab.py 3 6 "round-6 build=painter_amd/lib/libpainter_hip.so" "round-5 library=painter_amd/lib/libpainter_hip_base.so" > gpurun_out/libab6.log 2>&1; echo "libab6 rc=$?"; tail -4 gpurun_out/libab6.log ;;
    r6gemm)    timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -s -x -k "gemm256 or linear or vit_large or small_painter" > gpurun_out/r6gemm.log 2>&1; echo "r6gemm rc=$?"; grep -a "passed\|failed\|rror\|worst" gpurun_out/r6gemm.log | tail -12 ;;
    mixedab)   timeout 600 python tools/step_knob_ab.py 4 6 "full rounds + half tiles (round 6):12=0" "uniform tiles (round 5):12=1" > gpurun_out/mixedab.log 2>&1; echo "mixedab rc=$?"; tail -3 gpurun_out/mixedab.log ;;
    lntests)   timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "layernorm" > gpurun_out/lntests.log 2>&1; echo "lntests rc=$?"; tail -2 gpurun_out/lntests.log ;;
    conv2)     timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "conv or decoder or small_painter or vit_large_b8" > gpurun_out/conv2_tests.log 2>&1; echo "conv2 tests rc=$?"; tail -2 gpurun_out/conv2_tests.log
               timeout 300 python tools/conv_bench.py > gpurun_out/conv2_bench.log 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/conv2_bench.log ;;
    convab)    for l in libpainter_hip_prev.so libpainter_hip.so; do echo "== $l"; PAINTER_AMD_LIB=painter_amd/lib/$l timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/convab_bench.log 2>&1; cat gpurun_out/convab_bench.log
               timeout 900 python tools/step_lib_ab.py 3 6 "conv epilogues stored as whole pixel rows through LDS (+ weights two taps ahead)=painter_amd/lib/libpainter_hip.so" "committed=painter_amd/lib/libpainter_hip_prev.so" > gpurun_out/convab.log 2>&1; echo "convab rc=$?"; tail -3 gpurun_out/convab.log ;;
    patchsweep) for pt in 0 605 606 608 1204; do
                 (cd /tmp && for c in "FETCH_SIZE" "WRITE_SIZE"; do
                    PA_G256_PATCH=$pt PAINTER_AMD_SIDE_STREAM=0 timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OLDPWD/gpurun_out/pmc_$c -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --profile-steps 0 --min-seconds 0 > $OLDPWD/gpurun_out/pmc_$c.log 2>&1; done)
                 python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/traffic_patch_$pt.json > /dev/null 2>&1
                 rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
                 python -c "import json,sys; t=json.load(open('gpurun_out/traffic_patch_$pt.json')); print('patch', '$pt', {k: (v['hbm_read_bytes'] // 1000000, v['hbm_write_bytes'] // 1000000) for k, v in t.items() if not k.startswith('_')})"
               done 2>&1 | tee gpurun_out/patchsweep.log
               timeout 500 python tools/step_knob_ab.py 3 6 "patch 4 x 8 (shipped):11=0" "patch 6 x 5:11=605" "patch 6 x 6:11=606" > gpurun_out/patchab.log 2>&1; tail -4 gpurun_out/patchab.log ;;
    stepbounds) timeout 600 python tools/step_bounds.py > gpurun_out/stepbounds.log 2>&1; echo "stepbounds rc=$?"; grep -v amdgpu.ids gpurun_out/stepbounds.log ;;
    layoutprobe) timeout 300 python tools/gemm_layout_probe.py > gpurun_out/layoutprobe.log 2>&1; echo "layoutprobe rc=$?"; grep -v amdgpu.ids gpurun_out/layoutprobe.log ;;
    benchselftest) PAINTER_AMD_DDP_SELFTEST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-optimizer --no-reference-gpu --no-secondary --min-seconds 1 > gpurun_out/benchselftest.json 2> gpurun_out/benchselftest.err; echo "benchselftest rc=$?"; python -c "import json; ls_ = open('gpurun_out/benchselftest.json').read().strip().splitlines(); print('stdout lines', len(ls_), 'last is json', ls_[-1].startswith('{')); d = json.loads(ls_[-1]); print(d['value'], d['config']['rccl_ranks'], d['config']['grad_allreduce'][:40]); print(json.dumps(d['extra'], indent=0)[:1800])"; tail -3 gpurun_out/benchselftest.err ;;
    multitest) timeout 900 python -m pytest tests/test_parallel_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "bench_multi or selftest or ddp" > gpurun_out/multitest.log 2>&1; echo "multitest rc=$?"; tail -3 gpurun_out/multitest.log ;;
    mixedprobe) timeout 300 python tools/gemm_mixed_probe.py > gpurun_out/mixedprobe.log 2>&1; echo "mixedprobe rc=$?"; cat gpurun_out/mixedprobe.log ;;
    lnfwdab)   timeout 600 python tools/step_knob_ab.py 4 6 "LN forward, persistent waves (round 6):13=1" "one row per wave (round 5):13=0" > gpurun_out/lnfwdab.log 2>&1; echo "lnfwdab rc=$?"; tail -3 gpurun_out/lnfwdab.log ;;
    livey)     timeout 1200 python -m pytest tests/test_live_yardstick_gpu.py tests/test_model_gpu.py -m gpu -q -s -k "live_yardstick or h14" > gpurun_out/livey.log 2>&1; echo "livey rc=$?"; grep -v amdgpu.ids gpurun_out/livey.log | grep "head_dim 80\|h14.*bf16\|train_one_epoch at\|passed\|failed\|Error\|assert" | cut -c1-1500 ;;
    deltaprobe) timeout 300 python tools/attn_delta_probe.py > gpurun_out/deltaprobe.log 2>&1; echo "deltaprobe rc=$?"; cat gpurun_out/deltaprobe.log ;;
    *)         echo "unknown section $s" ;;
  esac
done
