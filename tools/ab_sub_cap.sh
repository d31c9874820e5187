#!/bin/bash
# Same-box A/B of the sub-batch cap (hooks build, OG_SUB_CAP) on the natural depth-32 statement at batch 4096 and 1024, and a
# check that the 2^18-wire headline keeps its plan.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; OUT=$PWD/gpurun_out; H=$PWD/owshen_amd/libowshen_gpu_hooks.so
run() { local name=$1 args=$2; shift 2
  env OWSHEN_GPU_LIB=$H "$@" timeout -s KILL 400 python bench.py $args --steps 3 --warmup 1 --no-cpu --no-legs --no-other > $OUT/cap_$name.json 2> $OUT/cap_$name.err
  python - $OUT/cap_$name.json "$name [$args] $*" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('---', sys.argv[2], '| proofs/s', d['value'], 'ms', d['ms_per_step'], 'plan', d['config']['sub_batch_plan']['sizes'], 'scratch GB', round(d['config']['hbm']['scratch_bytes_reserved']/2**30,1), d['repeatability'].get('summary'))
except Exception as e:
    print('---', sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
}
run n4096_cap256 "--natural --batch 4096" OG_SUB_CAP=256; run n4096_dflt "--natural --batch 4096" OG_X=0
run n4096_cap256b "--natural --batch 4096" OG_SUB_CAP=256; run n4096_dfltb "--natural --batch 4096" OG_X=0
run n1024_cap256 "--natural --batch 1024" OG_SUB_CAP=256; run n1024_dflt "--natural --batch 1024" OG_X=0
run n1024_cap256b "--natural --batch 1024" OG_SUB_CAP=256; run n1024_dfltb "--natural --batch 1024" OG_X=0
run dense_dflt "--batch 1024" OG_X=0
