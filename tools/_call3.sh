cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_msm26
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_msm26 -o r05 -- python $R/bench.py --workload msm26 --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_msm26.log 2>&1
f=$(find $R/gpurun_out/prof_msm26 -name "*kernel_stats*" | head -1); echo "stats: $f"; head -25 "$f" | cut -c1-200
cd $R
export VSTEPS=4
export VARIANTS=$'OG_X=0\nOG_MERGE_LH=0\nOWSHEN_GPU_LIB=/root/repo/owshen_amd/libowshen_gpu_minw3.so\nOG_X=0\nOG_MERGE_LH=0\nOWSHEN_GPU_LIB=/root/repo/owshen_amd/libowshen_gpu_minw3.so'
tools/gpu_round.sh r05c variants
timeout 900 python -m pytest tests/test_gpu_withdraw.py tests/test_plumbing.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -5
