#!/bin/bash
# One gpurun call, staged; every stage under its own timeout, logs under gpurun_out/.  usage: tools/gpu_round.sh <tag> [stages...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
REPO=$PWD
TAG=${1:-r04}; shift
stages="${@:-sanity tests bench prof}"
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name (timeout $to) $(date +%T)"; local t0=$(date +%s)
  timeout -s KILL $to "$@" > $OUT/$name.log 2>&1; RC=$?
  echo "=== $name rc=$RC $(( $(date +%s) - t0 ))s"; tail -n ${TAILN:-8} $OUT/$name.log | cut -c1-600
  return $RC
}
pmc_pass() { # name counters...
  local name=$1; shift
  ( cd /tmp; run pmc_$name ${PMC_TO:-300} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o $TAG -- python $REPO/bench.py ${PMC_ARGS:---batch 512 --steps 1 --warmup 0 --no-cpu --dense --no-legs} )
}
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    iso=d.get('stage_ms_per_step_isolated',{}); st=d.get('stage_ms_per_step',{})
    print('   value', d['value'], 'ms/step', d['ms_per_step'], 'pad', d['config'].get('padding','')[:6])
    print('   timed   ', {k: round(v,1) for k,v in st.items()})
    print('   isolated', {k: round(v,1) for k,v in iso.items()})
    for k in ('sparse_padding','dense_padding'):
        if k in d: print('  ',k, d[k]['value'], d[k]['ms_per_step'])
    if 'MSM' in d.get('metric',''): print('   msm stages', d.get('stage_ms_per_step'))
    for k in ('msm26','tree20'):
        if k in d: print('  ',k, d[k]['value'], d[k]['ms_per_step'], d[k].get('stage_ms_per_step'))
except Exception as e:
    print('   summary failed', e)
PY
}
for s in $stages; do
  case $s in
    sanity) run sanity 400 python -c "import time; t=time.time(); import torch; print('import torch', round(time.time()-t,1), 's', torch.cuda.get_device_name(0), torch.cuda.device_count()); import os; print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))" || exit 1
            echo "   cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) nproc: $(nproc) loadavg: $(cut -d' ' -f1-3 /proc/loadavg)" ;;
    tests) TAILN=25 run tests ${TESTS_TO:-1000} python -m pytest tests -x -q -m gpu --timeout=400 --durations=10 ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2; tail -n 1 $OUT/bench.log > $OUT/${TAG}_bench.json; summ $OUT/${TAG}_bench.json ;;
    bench_driver) run bench_driver 900 python bench.py --gpus 1 --steps 20 --warmup 5; tail -n 1 $OUT/bench_driver.log > $OUT/${TAG}_bench_driver.json; summ $OUT/${TAG}_bench_driver.json ;;
    variants)  # (OG_* switches exist only in the hooks build of the library: every variant, the empty one too, runs on it)
      i=0
      while IFS= read -r v; do
        [ -z "$v" ] && continue
        i=$((i+1))
        env OWSHEN_GPU_LIB=$REPO/owshen_amd/libowshen_gpu_hooks.so $v timeout -s KILL 300 python bench.py --steps ${VSTEPS:-3} --warmup 1 --no-cpu --no-legs ${VARGS:---dense} > $OUT/var_$i.log 2>&1
        echo "--- [$i] $v"; tail -n 1 $OUT/var_$i.log > $OUT/var_$i.json; summ $OUT/var_$i.json
      done <<< "$VARIANTS" ;;
    prof) ( cd /tmp; run prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --dense --no-legs --no-verify )
          find $OUT/prof -name "*kernel_stats*" | head -3 ;;
    prof_sparse) ( cd /tmp; run prof_sparse 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sparse -o $TAG -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --sparse --no-legs ) ;;
    pmc) # dense padding; ONE step of the batch-1024 headline under the prover's own sub-batch plan (round 4: the traffic bench.py
         # quotes is a reading of the very launches it times, not a per-point rescale)
         export PMC_ARGS="--batch 1024 --steps 1 --warmup 0 --no-cpu --dense --no-legs --no-isolated --no-verify"
         pmc_pass fetch FETCH_SIZE || { echo "pmc: first pass failed, skipping the rest"; unset PMC_ARGS; continue; }
         pmc_pass write WRITE_SIZE
         pmc_pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
         pmc_pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
         unset PMC_ARGS
         python tools/pmc_summary.py $TAG $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_sq > $OUT/${TAG}_pmc_summary.txt 2>$OUT/pmc_summary.err
         python tools/pmc_traffic.py $TAG dense $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_tcc --points 262144,262144,262137,131071 --windows 15,15,15,15 --total-proofs 1024 > $OUT/pmc_traffic_dense.log 2>&1
         cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
         tail -n 3 $OUT/pmc_summary.err; head -n 8 $OUT/${TAG}_pmc_summary.txt | cut -c1-300 ;;
    pmc_sparse) export PMC_ARGS="--batch 1024 --steps 1 --warmup 0 --no-cpu --sparse --no-legs --no-isolated --no-verify"
         pmc_pass sfetch FETCH_SIZE || { unset PMC_ARGS; continue; }
         pmc_pass swrite WRITE_SIZE
         pmc_pass ssq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
         pmc_pass stcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
         unset PMC_ARGS
         python tools/pmc_summary.py ${TAG}_sparse $OUT/pmc_sfetch $OUT/pmc_swrite $OUT/pmc_stcc $OUT/pmc_ssq > $OUT/${TAG}_sparse_pmc_summary.txt 2>>$OUT/pmc_summary.err
         python tools/pmc_traffic.py $TAG sparse $OUT/pmc_sfetch $OUT/pmc_swrite $OUT/pmc_ssq $OUT/pmc_stcc --points 131102,116699,262137,131071 --windows 16,16,15,15 --total-proofs 1024 > $OUT/pmc_traffic_sparse.log 2>&1
         cp profiles/pmc_traffic.json $OUT/pmc_traffic.json ;;
    msm26) run msm26 600 python bench.py --workload msm26 --steps 2 --warmup 1; tail -n 1 $OUT/msm26.log > $OUT/${TAG}_msm26.json; summ $OUT/${TAG}_msm26.json ;;
    tree20) run tree20 300 python bench.py --workload tree20 --steps 5 --warmup 1; tail -n 1 $OUT/tree20.log > $OUT/${TAG}_tree20.json ;;
    latency) run latency 300 python tools/latency.py; cp $OUT/latency.json $OUT/${TAG}_latency.json ;;
    bench_nat) run bench_nat 600 python bench.py --natural --batch 4096 --steps 2 --warmup 1; tail -n 1 $OUT/bench_nat.log > $OUT/${TAG}_bench_natural.json ;;
    multi_dry)
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_prove 300 python bench.py --gpus 2 --batch 64 --steps 1 --warmup 1 --no-cpu --no-other --no-legs
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_msm 300 python bench.py --gpus 2 --workload msm26 --log-n 20 --steps 1 --warmup 1 --no-cpu
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_tree 300 python bench.py --gpus 2 --workload tree20 --log-n 16 --steps 1 --warmup 1 --no-cpu ;;
    multi_dry6)  # round 6: the window-sharded prove mode with 2 ranks on this one GPU (gloo fallback: RCCL cannot start two ranks on
      # one device), the driver's launch form with ONE rank against the plain run, and the watchdog with a rank that never arrives
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_shard_windows 400 python bench.py --gpus 2 --shard windows --batch 16 --steps 2 --warmup 1 --no-cpu
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_shard_windows_nat 400 python bench.py --gpus 2 --shard windows --natural --batch 256 --steps 2 --warmup 1 --no-cpu
      run one_rank_torchrun 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --no-legs --dense --no-verify
      run one_rank_plain 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --no-legs --dense --no-verify
      OG_BENCH_OVERSUBSCRIBE=1 OG_BENCH_TEST_HANG_RANK=1 OG_BENCH_WATCHDOG_S=25 run watchdog_hung_rank 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --batch 64 --steps 1 --warmup 1 --no-cpu --no-other --no-legs
      { echo "# round 6 dry runs on ONE GPU (gpurun box): 2 ranks share the device, so the process group is gloo -- code-path evidence, not a measurement"
        for n in multi_dry_shard_windows multi_dry_shard_windows_nat one_rank_torchrun one_rank_plain watchdog_hung_rank; do echo "=== $n"; grep -E '^\{|WATCHDOG|self-launch|falling back|rank [0-9]+:' $OUT/$n.log | cut -c1-2500; done; } > $OUT/${TAG}_multi_dry_full.txt ;;
    prof_msm26) ( cd /tmp; run prof_msm26 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_msm26 -o $TAG -- python $REPO/bench.py --workload msm26 --steps 2 --warmup 1 --no-cpu --no-precomp )
          f=$(find $OUT/prof_msm26 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/msm26_trace.py "$f" > $OUT/${TAG}_msm26_trace.txt 2>&1; tail -n 40 $OUT/${TAG}_msm26_trace.txt ;;
    sysprobe) run sysprobe 60 bash -c 'for d in /sys/class/drm/card*/device; do echo "== $d"; grep PCI_SLOT $d/uevent; ls $d/hwmon/*/ 2>/dev/null | tr "\n" " "; echo; for f in $d/hwmon/*/freq1_input $d/hwmon/*/power1_average $d/hwmon/*/power1_input $d/hwmon/*/temp1_input; do [ -e $f ] && echo "$f = $(cat $f 2>&1)"; done; done; which rocm-smi amd-smi; rocm-smi -c -P --json 2>&1 | head -c 1500; echo; python -c "import torch; p=torch.cuda.get_device_properties(0); print(p); print([a for a in dir(p) if not a.startswith(\"_\")])"' ;;
    ab_rounds) TAILN=12 run ab_rounds ${AB_TO:-1500} tools/ab_binaries.sh run ${AB_TAGS:-r03 r04 HEAD} ${AB_ROUNDS:-2}; cp $OUT/ab_rounds.txt $OUT/${TAG}_ab_rounds.txt ;;
    coalescer) TAILN=60 run coalescer ${COAL_TO:-900} python tools/coalescer.py ${COAL_ARGS:-}; cp $OUT/coalescer.json $OUT/${TAG}_coalescer.json ;;
    msm26_ab)  # the lone MSM's first sort level: round 5's digit arrays + staged runs (default) against round 4's direct scatter, interleaved
      for v in new v1 new2 v1b; do
        case $v in v1*) e="OG_LONE_SORT_V1=1" ;; *) e="OG_X=0" ;; esac
        env OWSHEN_GPU_LIB=$REPO/owshen_amd/libowshen_gpu_hooks.so $e timeout -s KILL 300 python bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > $OUT/msm26_$v.json 2> $OUT/msm26_$v.err
        echo "--- msm26 $v ($e)"; summ $OUT/msm26_$v.json
      done ;;
    custom) run custom ${CUSTOM_TO:-600} bash -c "$CUSTOM" ;;
  esac
done
echo "=== done $(date +%T)"
