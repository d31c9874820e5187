cd $GRAFT_REPO_ROOT
H=$PWD/owshen_amd/libowshen_gpu_hooks.so
for v in p2 p4 p2b p4b; do
  case $v in p4*) e="OG_LONE_PARTS=4" ;; *) e="OG_X=0" ;; esac
  env OWSHEN_GPU_LIB=$H $e timeout -s KILL 300 python bench.py --workload msm26 --steps 4 --warmup 1 --no-cpu > gpurun_out/msm26d_$v.json 2> gpurun_out/msm26d_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/msm26d_{v}.json').read().strip().splitlines()[-1]); print(v, d['ms_per_step'], d['stage_ms_per_step'])
except Exception as e: print(v,'failed',e, open(f'gpurun_out/msm26d_{v}.err').read()[-600:])
PY
done
