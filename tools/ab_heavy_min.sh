cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")}, {k: (d[k]["accumulate_g1"], d[k]["heavy_g1"], d[k]["reduce_g1"]) for k in d if k.endswith("regions_ms")})'; }
for v in "OG_X=0" "OG_HEAVY=48" "OG_HEAVY=32" "OG_HEAVY=16" "OG_HEAVY=8" "OG_X=0"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,4 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
done
