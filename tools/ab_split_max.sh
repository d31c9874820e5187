#!/bin/bash
# A/B of the schedule of a handful of requests (hooks build): OG_SPLIT_MAX=1 -- only ONE request fans its queries out over the streams,
# 2..16 run as two half-batches side by side ("symmetric lanes") -- against OG_SPLIT_MAX=16: every call of <= 16 requests fans out.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")})'; }
for v in "OG_SPLIT_MAX=1" "OG_SPLIT_MAX=16" "OG_SPLIT_MAX=1" "OG_SPLIT_MAX=16"; do
  for mode in "--host-chains" ""; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 2,3,4,8,16 $mode > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural $mode: $(pr)"
  done
done
for v in "OG_SPLIT_MAX=1" "OG_SPLIT_MAX=16"; do
  for mode in "--host-chains" ""; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --sizes 2,4,8,16 $mode > gpurun_out/lat_ab.log 2>&1
    echo "--- $v 2^18 $mode: $(pr)"
  done
done
