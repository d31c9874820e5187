#!/bin/bash
# A/B of the assembly's four scalar products (hooks build): OG_ASM_W9_MAX=0 -- a lane per half-length chain (k_assemble_g1_muls_glv) --
# against a WAVE per chain with the group law in the wave-wide form (k_assemble_g1_muls_w9) for calls of up to 8 / 16 / 64 requests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")}, {k: d[k]["assemble"] for k in d if k.endswith("regions_ms")})'; }
timeout 900 python -m pytest tests/test_gpu_withdraw.py tests/test_gpu_groth16.py tests/test_gpu_multi.py tests/test_gpu_deposit.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
for v in "OG_ASM_W9_MAX=0" "OG_ASM_W9_MAX=16" "OG_ASM_W9_MAX=64" "OG_ASM_W9_MAX=0" "OG_ASM_W9_MAX=16" "OG_ASM_W9_MAX=64"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,2,4,8,16,32,64 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
done
for v in "OG_ASM_W9_MAX=0" "OG_ASM_W9_MAX=16"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --sizes 1,8 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v 2^18: $(pr)"
done
