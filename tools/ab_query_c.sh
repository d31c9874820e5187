#!/bin/bash
# the switches this script sets exist only in the hooks build of the library (owshen_amd/csrc/ctx.h, -DOG_AB_HOOKS)
export OWSHEN_GPU_LIB=${OWSHEN_GPU_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/owshen_amd/libowshen_gpu_hooks.so}
# same-box A/B of the proving-key query window (OG_QUERY_C = 16 | 17): dense headline, interleaved runs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for c in 16 17 16 17; do
  echo "== OG_QUERY_C=$c"
  OG_QUERY_C=$c timeout 300 python bench.py --steps 6 --warmup 2 --no-other --no-legs --no-cpu 2>/dev/null | tail -1 > gpurun_out/ab_c$c.json
  python - "$c" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/ab_c{sys.argv[1]}.json").read())
print(d["value"], d["ms_per_step"], d.get("stage_ms_per_step_isolated"))
P
done
