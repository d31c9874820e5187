#!/bin/bash
# One gpurun call: staged, every stage under its own timeout, logs under gpurun_out/.
# usage: tools/gpu_round.sh [stages...]   (default: all)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
stages="${@:-g2 tests bench_small bench prof}"
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name (timeout $to)"; local t0=$(date +%s)
  timeout $to "$@" > $OUT/$name.log 2>&1; local rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - t0 ))s"; tail -n 12 $OUT/$name.log
}
for s in $stages; do
  case $s in
    g2) run g2 200 python -m pytest tests/test_gpu_msm.py -x -q -k "g2" ;;
    tests) run tests 900 python -m pytest tests -x -q -m gpu --durations=8 ;;
    bench_small) run bench_small 400 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu ;;
    bench) run bench 900 python bench.py ;;
    bench_nat) run bench_nat 600 python bench.py --natural --no-cpu ;;
    prof) (cd /tmp && run prof 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r01 -- python $OLDPWD/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu); 
          find $OUT/prof -name "*stats*" | head ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
  esac
done
