#!/bin/bash
# Same-box A/B of two builds of the library: owshen_amd/libowshen_gpu.so (new) against owshen_amd/libowshen_gpu_ab.so (base,
# e.g. built from `git archive HEAD` with `make OUT=.../libowshen_gpu_ab.so BUILD=build_ab`).  Order: new, base, new.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ARGS="${AB_ARGS:---steps 2 --warmup 1 --no-cpu --no-dense}"
cp owshen_amd/libowshen_gpu.so /tmp/new.so
for t in new base new2; do
  case $t in base) cp owshen_amd/libowshen_gpu_ab.so owshen_amd/libowshen_gpu.so ;; *) cp /tmp/new.so owshen_amd/libowshen_gpu.so ;; esac
  timeout 300 python bench.py $ARGS > gpurun_out/ab_$t.json 2> gpurun_out/ab_$t.err || tail -5 gpurun_out/ab_$t.err
done
cp /tmp/new.so owshen_amd/libowshen_gpu.so
if [ -n "${AB_TESTS:-}" ]; then timeout 600 python -m pytest $AB_TESTS -m gpu -x -q 2>&1 | tail -3; fi
python - <<'PY'
import json
for t in ("new", "base", "new2"):
    try:
        d = json.loads(open(f"gpurun_out/ab_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], {k: round(v, 1) for k, v in d["stage_ms_per_step_isolated"].items()})
    except Exception as e:
        print(t, "failed", e)
PY
