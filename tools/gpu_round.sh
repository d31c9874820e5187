#!/bin/bash
# One gpurun call: staged, every stage under its own timeout, logs under gpurun_out/; a failed stage stops the
# round (GPU minutes are scarce).  usage: tools/gpu_round.sh [stages...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
REPO=$PWD
stages="${@:-probe tests bench_small bench prof}"
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name (timeout $to)"; local t0=$(date +%s)
  timeout -s KILL $to "$@" > $OUT/$name.log 2>&1; RC=$?
  echo "=== $name rc=$RC $(( $(date +%s) - t0 ))s"; tail -n 15 $OUT/$name.log | cut -c1-400
  return $RC
}
for s in $stages; do
  case $s in
    probe) run probe 240 python tools/gpu_probe_g2.py || exit 1 ;;
    tests) run tests 600 python -m pytest tests -x -q -m gpu --timeout=150 --durations=10 || exit 1 ;;
    bench_small) run bench_small 300 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu || exit 1 ;;
    bench)
      X=$(python -c "import json,sys; print(json.loads(open('$OUT/bench_small.log').read().strip().splitlines()[-1])['value'])" 2>/dev/null || echo 1000)
      B=$(python -c "print(min(1024, max(32, int($X * 40) // 32 * 32)))")
      echo "bench_small value=$X -> batch $B"
      run bench 420 python bench.py --batch $B --steps 2 --warmup 1 --cpu-seconds 12 || exit 1 ;;
    bench_nat) run bench_nat 600 python bench.py --natural --batch 4096 ;;
    prof) cd /tmp; run prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r01 -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu; cd $REPO
          find $OUT/prof -name "*stats*" | head ;;
    pmc) cd /tmp
         run pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r01 -- python $REPO/bench.py --batch 28 --steps 1 --warmup 0 --no-cpu
         run pmc_write 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r01 -- python $REPO/bench.py --batch 28 --steps 1 --warmup 0 --no-cpu
         cd $REPO; find $OUT/pmc_fetch $OUT/pmc_write -name "*.csv" | head ;;
    variants)
      for v in ${VARIANTS:-"A=0" "OG_NO_ORDER=1" "OG_ACC_MINW=2"}; do
        n=$(echo $v | tr -c 'A-Za-z0-9' '_')
        env $v timeout -s KILL 200 python bench.py --batch ${VBATCH:-256} --steps 1 --warmup 1 --no-cpu > $OUT/var_$n.log 2>&1
        echo "--- $v"; tail -1 $OUT/var_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'])" 2>&1 | cut -c1-400
      done ;;
    mulmod) run mulmod 200 python tools/gpu_probe.py; python -c "
import json; d=json.load(open('$OUT/probe.json')); print({k:round(v['mulmod_per_s']/1e9,1) for k,v in d['mulmod'].items()}); print({k:round(v['hash_per_s']/1e6,2) for k,v in d.items() if k.startswith('mimc7')})" ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1 ;;
  esac
done
