#!/bin/bash
# the switches this script sets exist only in the hooks build of the library (owshen_amd/csrc/ctx.h, -DOG_AB_HOOKS)
export OWSHEN_GPU_LIB=${OWSHEN_GPU_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/owshen_amd/libowshen_gpu_hooks.so}
# same-box A/B of the bucket-reduction variants (dense headline, interleaved, two rounds):
#   base | OG_RED_ALT=1 (2 waves / SIMD builds) | SEG = 16 build | 17-bit windows with each
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-other --no-legs --no-cpu 2>/dev/null | tail -1 > gpurun_out/abr_$name.json
  python - "$name" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/abr_{sys.argv[1]}.json").read())
s = d.get("stage_ms_per_step_isolated", {})
print(f"{sys.argv[1]:>14} {d['value']:8.2f} proofs/s  {d['ms_per_step']:8.1f} ms   isolated acc {s.get('accumulate_g1',0)+s.get('accumulate_g2',0):7.1f}  reduce {s.get('reduce_g1',0):5.1f} + {s.get('reduce_g2',0):5.1f}")
P
}
SEG16=$PWD/owshen_amd/libowshen_gpu_seg16.so
for round in 1 2; do
  run base OG_X=0
  run red_alt OG_RED_ALT=1
  run seg16 OWSHEN_GPU_LIB=$SEG16
  run c17 OG_QUERY_C=17
  run c17_red_alt OG_QUERY_C=17 OG_RED_ALT=1
  run c17_seg16 OG_QUERY_C=17 OWSHEN_GPU_LIB=$SEG16
done
