#!/bin/bash
# A/B of the witness form for calls of a few requests (hooks build): OG_WITNESS_W9=0 -- the lane-local kernels (wave per proof up to 16
# requests, a lane pair per proof above) -- against the wave-wide form (k_w9_*: one permutation per wave, three launches).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")}, {k: d[k]["witness"] for k in d if k.endswith("regions_ms")})'; }
for v in "OG_WITNESS_W9=0" "OG_WITNESS_W9=1" "OG_WITNESS_W9=0" "OG_WITNESS_W9=1"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,2,8,16,64,128,256,512 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
done
for v in "OG_WITNESS_W9=0" "OG_WITNESS_W9=1"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --sizes 1,8,64 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v 2^18: $(pr)"
done
