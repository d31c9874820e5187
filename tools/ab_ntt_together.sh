#!/bin/bash
# A/B of the quotient's passes for calls that are waited for (hooks build): OG_NTT_TOGETHER_MAX=0 -- a, b, c one after the other, 11
# launches at domain 2^15 -- against the default -- every pass over the three arrays at once (gridDim.z = 3), 6 launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")}, {k: d[k]["h_poly"] for k in d if k.endswith("regions_ms")})'; }
timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_groth16.py tests/test_gpu_withdraw.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
for v in "OG_NTT_TOGETHER_MAX=0" "OG_X=1" "OG_NTT_TOGETHER_MAX=0" "OG_X=1"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,2,8,16,64,256 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --sizes 1,8,64 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v 2^18: $(pr)"
done
