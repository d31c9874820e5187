cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; H=$PWD/owshen_amd/libowshen_gpu_hooks.so
for r in 1 2; do for m in 1024 2048 4096 8192; do
  echo -n "OG_MIMC_W9_MAX=$m tree20 ms: "; env OWSHEN_GPU_LIB=$H OG_MIMC_W9_MAX=$m timeout 200 python bench.py --workload tree20 --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'
done; done
