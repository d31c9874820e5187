#!/bin/bash
# the switches this script sets exist only in the hooks build of the library (owshen_amd/csrc/ctx.h, -DOG_AB_HOOKS)
export OWSHEN_GPU_LIB=${OWSHEN_GPU_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/owshen_amd/libowshen_gpu_hooks.so}
# lone-MSM first-level chunk size A/B (round 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for c in ${CHUNKS:-32768 131072 32768 262144 65536}; do
  OG_LONE_CHUNK=$c timeout 300 python bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > gpurun_out/msm26_c$c.json 2>/dev/null
  python - "$c" <<'PY'
import json,sys
c=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/msm26_c{c}.json').read().strip().splitlines()[-1]); print('chunk', c, d['ms_per_step'], d['stage_ms_per_step'])
except Exception as e: print(c,'failed',e)
PY
done
