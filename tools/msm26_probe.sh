#!/bin/bash
# the switches this script sets exist only in the hooks build of the library (owshen_amd/csrc/ctx.h, -DOG_AB_HOOKS)
export OWSHEN_GPU_LIB=${OWSHEN_GPU_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/owshen_amd/libowshen_gpu_hooks.so}
# msm26 plain-bases variants on one box (round 4): piece counts, then a kernel trace of the default
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { # tag env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > gpurun_out/msm26_$tag.json 2> gpurun_out/msm26_$tag.err || tail -3 gpurun_out/msm26_$tag.err
  python - "$tag" <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/msm26_{t}.json').read().strip().splitlines()[-1]); print(t, d['ms_per_step'], d['stage_ms_per_step'])
except Exception as e: print(t,'failed',e)
PY
}
for v in ${PIECES:-1 8 4 16}; do one p$v OG_LONE_PIECES=$v; done
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_msm26 -o r04 -- python $OLDPWD/bench.py --workload msm26 --steps 2 --warmup 1 --no-cpu > $OLDPWD/gpurun_out/prof_msm26.log 2>&1 )
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_msm26/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:14]:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.2f'%(float(r['AverageNs'])/1e6)).rjust(9))
PY
