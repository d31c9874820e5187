#!/bin/bash
# A/B of the fanned-out call's assembly (hooks build), same box, interleaved:
#   old:  OG_ASM_EARLY=0 OG_ASM_G2_TREE=0 -- the G1 products, sums and both inversions behind the H query on stream 0; B's half a lane
#         per proof, 65 additions deep
#   new:  the defaults -- the G1 products and A on a side stream beside the quotient and the H query, C = L + H + three products behind
#         them; B's half a wave per proof, the 64 table entries of s delta2 as a tree (8 additions deep)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")})'; }
timeout 900 python -m pytest tests/test_gpu_withdraw.py tests/test_gpu_groth16.py tests/test_gpu_fullsize_pipeline.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3
for v in "OG_ASM_EARLY=0 OG_ASM_G2_TREE=0" "OG_ASM_EARLY=1 OG_ASM_G2_TREE=0" "OG_ASM_EARLY=1 OG_ASM_G2_TREE=1" "OG_ASM_EARLY=0 OG_ASM_G2_TREE=0" "OG_ASM_EARLY=1 OG_ASM_G2_TREE=0" "OG_ASM_EARLY=1 OG_ASM_G2_TREE=1"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,2,8,16,64,256,1024 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --sizes 1,8,64 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v 2^18: $(pr)"
done
