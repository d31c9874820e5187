#!/bin/bash
# A/B of where a one-sub-batch call's witnesses are generated (hooks build): OG_GEN_ONE_SUB=0 -- a slab up front, the host waiting for
# the walk before it issues the queries' ~80 launches -- against the default -- inside prove_enqueue, everything queued while the walk runs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")})'; }
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
for v in "OG_GEN_ONE_SUB=0" "OG_X=1" "OG_GEN_ONE_SUB=0" "OG_X=1"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,2,8,16,64,256,1024 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --sizes 1,8,64 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v 2^18: $(pr)"
done
