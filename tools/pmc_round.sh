#!/bin/bash
# rocprofv3 PMC passes over the headline bench command (separate passes per counter group, kernel-trace only:
# no sys/hip/hsa tracing next to --pmc).  usage: tools/pmc_round.sh <tag> [bench args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=${1:-r02}; shift
ARGS="${@:---batch 1024 --steps 1 --warmup 0 --no-cpu}"
export TMPDIR=/tmp
cd /tmp
pass() { # name counters...
  local name=$1; shift
  echo "=== pmc $name: $*"; local t0=$(date +%s)
  timeout -s KILL 420 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o $TAG -- python $REPO/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
  echo "=== pmc $name rc=$? $(( $(date +%s) - t0 ))s"; tail -n 2 $OUT/pmc_$name.log | cut -c1-300
}
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
cd $REPO
python tools/pmc_summary.py $TAG $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_sq > $OUT/${TAG}_pmc_summary.txt 2>$OUT/pmc_summary.err
tail -n 5 $OUT/pmc_summary.err; head -n 30 $OUT/${TAG}_pmc_summary.txt | cut -c1-250
