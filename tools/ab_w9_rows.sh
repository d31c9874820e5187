#!/bin/bash
# A/B of the wave-wide MiMC7 round (hooks build): OG_W9_ROWS=0 -- four products deep in one row of the wave (as first built) --
# against OG_W9_ROWS=1 -- three deep over two rows (t^4 beside t^3, t^6 beside t^7: mimc7.hip.h w9_mimc7_round) -- and 2, the default:
# the same with the 32-bit Montgomery digit.
# The witness walk of a handful of requests, the 2^20-leaf tree, one leaf appended; same box, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
H=$R/owshen_amd/libowshen_gpu_hooks.so
pr() { tail -1 gpurun_out/lat_ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k]["median_ms"] for k in d if k.startswith("batch_") and not k.endswith("_ms")}, {k: d[k]["witness"] for k in d if k.endswith("regions_ms")})'; }
timeout 600 python -m pytest tests/test_gpu_tree.py tests/test_gpu_withdraw.py tests/test_gpu_field_mimc7.py -x -q -m gpu 2>&1 | tail -4
python tools/w9_round_probe.py
for v in "OG_W9_ROWS=0" "OG_W9_ROWS=1" "OG_W9_ROWS=2" "OG_W9_ROWS=0" "OG_W9_ROWS=1" "OG_W9_ROWS=2"; do
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/latency.py --natural --sizes 1,8,64,512 > gpurun_out/lat_ab.log 2>&1
    echo "--- $v natural: $(pr)"
    env OWSHEN_GPU_LIB=$H $v timeout 200 python bench.py --workload tree20 --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("    tree20 ms", d["ms_per_step"])'
    env OWSHEN_GPU_LIB=$H $v timeout 200 python tools/append_latency.py > /dev/null 2>&1; python -c "import json; d=json.load(open(\"gpurun_out/append_latency.json\")); print(\"    append\", {k: v[\"kernels\"][\"median_ms\"] for k, v in d[\"rows\"].items()})"
done
