#!/bin/bash
# Same-box A/B of the query window sizes on the natural depth-32 statement (hooks build): 16 bits everywhere (OG_QUERY_C=16),
# 15 bits everywhere (OG_QUERY_C=15), the default by size (B 13, A / L / H 15); throughput at batch 4096 and the latency of 1 / 8 / 64 requests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; OUT=$PWD/gpurun_out; H=$PWD/owshen_amd/libowshen_gpu_hooks.so
run() { local name=$1; shift
  env OWSHEN_GPU_LIB=$H "$@" timeout -s KILL 400 python bench.py --natural --batch 4096 --steps 3 --warmup 1 --no-cpu --no-legs > $OUT/c15_$name.json 2> $OUT/c15_$name.err
  python - $OUT/c15_$name.json "$name $*" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('---', sys.argv[2], '| proofs/s', d['value'], 'ms', d['ms_per_step'], 'windows', d['config'].get('query_window_bits'), 'verified', d['repeatability'].get('summary'))
    print('      iso', {k: round(v,1) for k,v in (d.get('stage_ms_per_step_isolated') or {}).items()})
except Exception as e:
    print('---', sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
lat() { local name=$1; shift
  env OWSHEN_GPU_LIB=$H "$@" timeout -s KILL 300 python tools/latency.py --natural > $OUT/lat_$name.log 2>&1
  python - "$name $*" <<'PY'
import json,sys
d=json.load(open('gpurun_out/latency_natural.json'))
print('--- latency', sys.argv[1], {k: (v['median_ms'], v['min_ms']) for k,v in d.items() if k.startswith('batch_') and isinstance(v, dict) and 'median_ms' in v}, {k: round(v,2) for k,v in d.get('batch_1_regions_ms',{}).items()})
PY
}
for s in ${STAGES:-thr lat}; do case $s in
  thr) run c16_a OG_QUERY_C=16; run c15_a OG_QUERY_C=15; run dflt_a OG_X=0; run c16_b OG_QUERY_C=16; run c15_b OG_QUERY_C=15; run dflt_b OG_X=0 ;;
  lat) lat c16 OG_QUERY_C=16; lat c15 OG_QUERY_C=15; lat dflt OG_X=0; lat c16b OG_QUERY_C=16; lat dfltb OG_X=0 ;;
esac; done
