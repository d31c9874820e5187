cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
H=$PWD/owshen_amd/libowshen_gpu_hooks.so
for v in new v1 seg8k seg16k new2; do
  case $v in v1) e="OG_LONE_SORT_V1=1" ;; seg8k) e="OG_HEAVY_SEG=8192" ;; seg16k) e="OG_HEAVY_SEG=16384" ;; *) e="OG_X=0" ;; esac
  env OWSHEN_GPU_LIB=$H $e timeout -s KILL 300 python bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > gpurun_out/msm26b_$v.json 2> gpurun_out/msm26b_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/msm26b_{v}.json').read().strip().splitlines()[-1]); print(v, d['ms_per_step'], d['stage_ms_per_step'])
except Exception as e: print(v,'failed',e)
PY
done
( cd /tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_msm26b $GRAFT_REPO_ROOT/gpurun_out/prof_tree20
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_msm26b -o r05 -- python $GRAFT_REPO_ROOT/bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_msm26b.log 2>&1
  grep -E "k_lone|k_sub|k_accumulate|k_scan_sl" $GRAFT_REPO_ROOT/gpurun_out/prof_msm26b/r05_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tree20 -o r05 -- python $GRAFT_REPO_ROOT/bench.py --workload tree20 --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_tree20.log 2>&1
  head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_tree20/r05_kernel_stats.csv | cut -c1-200 )
timeout 300 python bench.py --in-process --gpus 1 --steps 3 --warmup 1 --batch 1024 > gpurun_out/inprocess1.json 2> gpurun_out/inprocess1.err; tail -c 1500 gpurun_out/inprocess1.json; tail -3 gpurun_out/inprocess1.err
timeout 900 python -m pytest tests/test_plumbing.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -4
