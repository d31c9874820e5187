#!/bin/bash
# A/B of the lone 2^26 MSM's window size on one box (hooks build): 16-bit windows (OG_LONE_C=16) against 20-bit windows (default
# from 2^24 points on), interleaved; then piece counts of the carried sweep, the single launch set, and a kernel-stats profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; OUT=$PWD/gpurun_out; REPO=$PWD; export TMPDIR=/tmp
H=$REPO/owshen_amd/libowshen_gpu_hooks.so
run() { # name env...
  local name=$1; shift
  env OWSHEN_GPU_LIB=$H "$@" timeout -s KILL 300 python bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > $OUT/c20_$name.json 2> $OUT/c20_$name.err
  python - $OUT/c20_$name.json "$name $*" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('---', sys.argv[2], '| ms', d['ms_per_step'], 'c', d['config'].get('window_bits'), 'stages', d.get('stage_ms_per_step'), 'KAT', (d['config'].get('known_answer') or '')[:30])
except Exception as e:
    print('---', sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
for s in ${STAGES:-ab pieces prof}; do case $s in
  ab) run c16_a OG_LONE_C=16; run c20_a OG_X=0; run c16_b OG_LONE_C=16; run c20_b OG_X=0 ;;
  pieces) run c20_p8 OG_LONE_PIECES=8; run c20_p32 OG_LONE_PIECES=32; run c20_single OG_LONE_HALVES=0; run c20_nosweep OG_LONE_SWEEP=0 ;;
  prof) ( cd /tmp; timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c20 -o c20 -- python $REPO/bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > $OUT/prof_c20.log 2>&1 )
        f=$(find $OUT/prof_c20 -name "*kernel_stats*" | head -1); [ -n "$f" ] && { cp $f $OUT/r05_kernel_stats_msm26_c20.csv; head -14 $f | cut -c1-160; } ;;
esac; done
