#!/bin/bash
# Runs a sequence of GPU steps on the gpurun box, each under its own timeout; a step that fails
# an assertion does not stop the sequence, a step that TIMES OUT does (no GPU work after a hang).
# usage: tools/gpu_round.sh step1 step2 ...   (steps: see the case list; ABLIBS='a b' ... ablibs benches ptina_amd/libmiptina_a.so, _b.so)
mkdir -p gpurun_out
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-12}   # also for the rocprofv3 steps, whose tool initialises HIP before libmiptina loads
run() {  # name seconds command...
  local name=$1 secs=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/round.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/round.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/round.log; exit 1; fi
}
: > gpurun_out/round.log
for step in "$@"; do
  case $step in
    smoke)       run smoke 300 python __graft_entry__.py smoke ;;
    soak)        run soak 1000 python tools/soak.py ${SOAK_LAUNCHES:-20000} ;;
    tests_soak)  run tests_soak 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k 'finalisation or hint or film_api' ;;
    diag)        run diag 600 python tools/gpu_diag.py parity timing ;;
    diag_parity) run diag_parity 400 python tools/gpu_diag.py parity ;;
    diag_timing) run diag_timing 500 python tools/gpu_diag.py timing ;;
    ab)          run ab 120 python tools/gpu_diag.py ab ;;
    tree)        run tree 300 python tools/gpu_diag.py tree ;;
    big)         run big 900 python tools/gpu_diag.py big ;;
    util)        run util 300 python tools/gpu_diag.py util ;;
    slabs)       run slabs 300 python tools/gpu_diag.py slabs ;;
    blk)         run blk 300 python tools/gpu_diag.py blk ;;
    pipe)        run pipe 400 python tools/gpu_diag.py pipe ;;
    timeline)    run timeline 300 python tools/gpu_diag.py timeline ;;
    fixedcost)   run fixedcost 300 python tools/gpu_diag.py fixedcost ;;
    timeline2)   MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_tl2.so run timeline2 300 python tools/gpu_diag.py timeline ;;   # needs tools/ab_build.sh tl2 "-DMPT_X_TIMELINE2=1"
    c3)          run c3 300 python tools/gpu_diag.py c3 ;;
    configs)     run configs 600 python tools/run_configs.py ;;
    big_ab)      MIPTINA_WIDE=0 run big_bin 400 python tools/run_configs.py C4 C5; MIPTINA_WIDE=1 run big_wide 400 python tools/run_configs.py C4 C5 ;;
    c5sah)       run c5sah 600 python tools/gpu_diag.py c5sah ;;
    probe)       run probe 300 python tools/gpu_diag.py probe ;;
    readback)    run readback 200 python tools/gpu_diag.py readback ;;
    stamps)      MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_stamp.so run stamps 200 python tools/gpu_diag.py stamps ;;   # needs a -DMPT_X_STAMPS=1 build under that name
    stamps_big)  for S in c4 c5; do STAMP_SCENE=$S MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_stamp.so run stamps_$S 300 python tools/gpu_diag.py stamps; done ;;
    shares_sync) run shares_sync 300 python tools/gpu_diag.py shares_sync ;;
    sync_sweep)  run sync_sweep 300 python tools/gpu_diag.py sync_sweep ;;
    ubench)      run ubench 200 tools/microbench/valu_microbench --json ;;
    l4ab)        # the LDS-resident 4-wide kernel of each library in ABLIBS (main = the product library), stamps for *stamp
                 for L in $ABLIBS; do
                   if [ "$L" = main ]; then run l4_$L 200 python tools/ab_lds4.py 0 1
                   elif [[ "$L" == *stamp* ]]; then MIPTINA_OPTS=lds_wide=1 MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_$L.so run l4_$L 200 python tools/gpu_diag.py stamps
                   else MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_$L.so run l4_$L 200 python tools/ab_lds4.py 1; fi
                 done ;;
    gbench)      run gbench 400 tools/microbench/gather_microbench ;;
    ubench_pmc)  run ubench_pmc 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/ubench_pmc -- tools/microbench/valu_microbench ;;
    bench_quick) run bench_quick 300 python bench.py --no-pmc --no-cpu-baseline --no-configs ;;
    bench_nofin) MIPTINA_OPTS=finalise=0 run bench_nofin 300 python bench.py --no-pmc --no-cpu-baseline --no-configs ;;
    tests_fin)   run tests_fin 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k 'finalisation or hint or film_api or batching or full_size or smoke or pipelin' ;;
    ablibs)      n=0; for L in $ABLIBS; do n=$((n+1)); O=$MIPTINA_OPTS; case $L in base*|nofin*|*nofin) O=finalise=0 ;; *noimg) O=finalise=2 ;; esac;   # (builds without the in-kernel finalisation need the combine pass)
                   F=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_$L.so; [ "$L" = main ] && F=$GRAFT_REPO_ROOT/ptina_amd/libmiptina.so;
                   MIPTINA_OPTS=$O MIPTINA_LIB=$F run ab${n}_$L 300 python bench.py --no-pmc --no-cpu-baseline --no-configs; done ;;
    tests_fast)  run tests_fast 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k 'fast_build or strict_build or full_size or batching or pipelining or lds_and_gather or lobes or ordered or idle or work_item or quantised' ;;
    tests_big)   run tests_big 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k 'config5 or config4 or mid_size or large_scene or lds_and_gather or where_the_tree' ;;
    big_wide)    MIPTINA_WIDE=1 run big_wide 400 python tools/run_configs.py C4 C5 ;;
    big_oct)     MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_oct.so MIPTINA_OPTS=wide8=1 CONFIGS_OUT=configs_oct.json run big_oct 500 python tools/run_configs.py C4 C5 ;;
    tests_oct)   run tests_oct 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k 'octant or random_scenes' ;;
    big_ablibs)  for L in $ABLIBS; do MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_$L.so run big_$L 400 python tools/run_configs.py C4 C5; done ;;
    tests)       run tests 900 python -m pytest tests -m gpu -x -q ;;
    tests_all)   run tests_all 900 python -m pytest tests -m gpu -q ;;
    bench)       run bench 600 python bench.py ;;
    prof)        (cd /tmp; run_dir=$GRAFT_REPO_ROOT/gpurun_out/prof; rm -rf $run_dir; mkdir -p $run_dir;
                  cd $GRAFT_REPO_ROOT;
                  run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-configs) ;;
    hiptrace)    (run_dir=$GRAFT_REPO_ROOT/gpurun_out/hiptrace; rm -rf $run_dir; mkdir -p $run_dir;
                  run hiptrace 400 rocprofv3 --hip-trace --kernel-trace --output-format csv -d gpurun_out/hiptrace -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-pmc --no-configs) ;;
    slabtrace)   (run_dir=$GRAFT_REPO_ROOT/gpurun_out/slabtrace; rm -rf $run_dir; mkdir -p $run_dir;
                  STRIPE=16 run slabtrace 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/slabtrace -- python3 tools/slab_trace.py 8 3 40) ;;
    counters)    rocprofv3 -L > gpurun_out/counters_list.txt 2>&1; echo "counters listed" ;;
    pmc1)        run pmc1 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --role pmc-child --steps 3 --warmup 1 ;;
    pmc2)        run pmc2 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --role pmc-child --steps 3 --warmup 1 ;;
    pmc3)        run pmc3 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc3 -- python3 bench.py --role pmc-child --steps 3 --warmup 1 ;;
    pmc4)        run pmc4 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc4 -- python3 bench.py --role pmc-child --steps 3 --warmup 1 ;;
    pmc5)        run pmc5 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --output-format csv -d gpurun_out/pmc5 -- python3 bench.py --role pmc-child --steps 3 --warmup 1 ;;
    pmcbig1)     run pmcbig1 500 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcbig1 -- python3 tools/run_configs.py C4 C5 ;;
    pmcbig2)     run pmcbig2 500 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmcbig2 -- python3 tools/run_configs.py C4 C5 ;;
    pmcbig3)     run pmcbig3 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcbig3 -- python3 tools/run_configs.py C4 C5 ;;
    pmcbig4)     run pmcbig4 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmcbig4 -- python3 tools/run_configs.py C4 C5 ;;
    pmcbig5)     run pmcbig5 500 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --output-format csv -d gpurun_out/pmcbig5 -- python3 tools/run_configs.py C4 C5 ;;
    pmcoct3)     MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_oct.so MIPTINA_OPTS=wide8=1 run pmcoct3 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcoct3 -- python3 tools/run_configs.py C4 C5 ;;
    pmcoct4)     MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_oct.so MIPTINA_OPTS=wide8=1 run pmcoct4 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmcoct4 -- python3 tools/run_configs.py C4 C5 ;;
    pmcoct1)     MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_oct.so MIPTINA_OPTS=wide8=1 run pmcoct1 500 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcoct1 -- python3 tools/run_configs.py C4 C5 ;;
    # (TA_* / TCP_* counter passes over run_configs.py hung rocprofv3 on this pool -- 7 minutes without output -- and are not offered)
    torchrun1)   run torchrun1 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --force-comm --no-cpu-baseline --no-pmc ;;
    *) echo "unknown step $step" ;;
  esac
done
exit 0
