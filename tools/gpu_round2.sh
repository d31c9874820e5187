#!/bin/bash
# One gpurun call, staged; every stage under its own timeout, logs under gpurun_out/.  usage: tools/gpu_round2.sh <tag> [stages...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
REPO=$PWD
TAG=${1:-r02}; shift
stages="${@:-sanity probe tests bench prof pmc}"
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name (timeout $to) $(date +%T)"; local t0=$(date +%s)
  timeout -s KILL $to "$@" > $OUT/$name.log 2>&1; RC=$?
  echo "=== $name rc=$RC $(( $(date +%s) - t0 ))s"; tail -n ${TAILN:-8} $OUT/$name.log | cut -c1-600
  return $RC
}
pmc_pass() { # name counters...
  local name=$1; shift
  ( cd /tmp; run pmc_$name ${PMC_TO:-240} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o $TAG -- python $REPO/bench.py ${PMC_ARGS:---batch 512 --steps 1 --warmup 0 --no-cpu --no-dense} )
}
for s in $stages; do
  case $s in
    sanity) run sanity 400 python -c "import time; t=time.time(); import torch; print('import torch', round(time.time()-t,1), 's', torch.cuda.get_device_name(0), torch.cuda.device_count()); import os; print('cpus', os.cpu_count())" || exit 1 ;;
    probe) run probe 300 python tools/gpu_probe.py; cp $OUT/probe.json $OUT/${TAG}_probe.json 2>/dev/null ;;
    tests) TAILN=25 run tests ${TESTS_TO:-1000} python -m pytest tests -x -q -m gpu --timeout=400 --durations=15 ;;
    newtests) TAILN=25 run newtests 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_multi.py tests/test_gpu_withdraw.py tests/test_gpu_msm.py -x -q -m gpu --timeout=400 --durations=15 ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 600 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1; tail -n 1 $OUT/bench.log > $OUT/${TAG}_bench.json ;;
    bench_quick) run bench_quick 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-dense; tail -n 1 $OUT/bench_quick.log > $OUT/${TAG}_bench_quick.json ;;
    counters) ( cd /tmp; rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]+|GRBM_[A-Z_]+|TCP_[A-Z0-9_]+)\b" | sort -u | tr '\n' ' ' > $OUT/${TAG}_counters.txt ); wc -c $OUT/${TAG}_counters.txt ;;
    pmc_icache) pmc_pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY
                python tools/pmc_summary.py ${TAG}_icache $OUT/pmc_icache > $OUT/${TAG}_icache_summary.txt 2>>$OUT/pmc_summary.err; head -n 8 $OUT/${TAG}_icache_summary.txt | cut -c1-300 ;;
    pmc_sq) pmc_pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
            pmc_pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE ;;
    variants)
      for v in ${VARIANTS:-"A=0"}; do
        n=$(echo $v | tr -c 'A-Za-z0-9' '_')
        env $v timeout -s KILL 200 python bench.py --batch ${VBATCH:-1024} --steps 2 --warmup 1 --no-cpu --no-dense > $OUT/var_$n.log 2>&1
        echo "--- $v"; tail -1 $OUT/var_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step_isolated'])" 2>&1 | cut -c1-400
      done ;;
    multi_dry) # two ranks on the one GPU of this box: exercises the self-launch, the rank checks and the collectives' code paths
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_prove 300 python bench.py --gpus 2 --batch 64 --steps 1 --warmup 1 --no-cpu --no-dense
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_msm 300 python bench.py --gpus 2 --workload msm26 --log-n 20 --steps 1 --warmup 1 --no-cpu
      OG_BENCH_OVERSUBSCRIBE=1 run multi_dry_tree 300 python bench.py --gpus 2 --workload tree20 --log-n 16 --steps 1 --warmup 1 --no-cpu
      run multi_refuse 120 python bench.py --gpus 2 --batch 64 --steps 1; echo "(expected: refusal, rc != 0)" ;;
    latency) run latency 300 python tools/latency.py; cp $OUT/latency.json $OUT/${TAG}_latency.json ;;
    bench_nat) run bench_nat 600 python bench.py --natural --batch 4096 --steps 2 --warmup 1; tail -n 1 $OUT/bench_nat.log > $OUT/${TAG}_bench_natural.json ;;
    bench_legacy) OG_SORT_LEGACY=1 run bench_legacy 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-dense ;;
    msm26) run msm26 600 python bench.py --workload msm26 --steps 2 --warmup 1; tail -n 1 $OUT/msm26.log > $OUT/${TAG}_msm26.json ;;
    tree20) run tree20 300 python bench.py --workload tree20 --steps 5 --warmup 1; tail -n 1 $OUT/tree20.log > $OUT/${TAG}_tree20.json ;;
    prof) ( cd /tmp; run prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --no-dense )
          find $OUT/prof -name "*kernel_stats*" | head -3 ;;
    pmc) pmc_pass fetch FETCH_SIZE || { echo "pmc: first pass failed, skipping the rest"; continue; }
         pmc_pass write WRITE_SIZE
         pmc_pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
         pmc_pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
         python tools/pmc_summary.py $TAG $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_sq > $OUT/${TAG}_pmc_summary.txt 2>$OUT/pmc_summary.err
         tail -n 3 $OUT/pmc_summary.err; head -n 12 $OUT/${TAG}_pmc_summary.txt | cut -c1-300 ;;
    pmc_dense) PMC_ARGS="--batch 512 --steps 1 --warmup 0 --no-cpu --dense" 
         export PMC_ARGS
         pmc_pass dfetch FETCH_SIZE || continue
         pmc_pass dwrite WRITE_SIZE
         python tools/pmc_summary.py ${TAG}_dense $OUT/pmc_dfetch $OUT/pmc_dwrite > $OUT/${TAG}_dense_pmc_summary.txt 2>>$OUT/pmc_summary.err ;;
  esac
done
echo "=== done $(date +%T)"
