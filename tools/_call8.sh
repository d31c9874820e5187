cd $GRAFT_REPO_ROOT
H=$PWD/owshen_amd/libowshen_gpu_hooks.so
for v in halves single halves2 single2; do
  case $v in single*) e="OG_LONE_HALVES=0" ;; *) e="OG_X=0" ;; esac
  env OWSHEN_GPU_LIB=$H $e timeout -s KILL 300 python bench.py --workload msm26 --steps 4 --warmup 1 --no-cpu > gpurun_out/msm26c_$v.json 2> gpurun_out/msm26c_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/msm26c_{v}.json').read().strip().splitlines()[-1]); print(v, d['ms_per_step'], d['stage_ms_per_step'], d['config']['known_answer'][:40])
except Exception as e: print(v,'failed',e, open(f'gpurun_out/msm26c_{v}.err').read()[-600:])
PY
done
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_parity_r2.py -m gpu -x -q -k "msm" 2>&1 | tail -4
