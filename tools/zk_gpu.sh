cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/zkprof -o r06g -- python $R/tools/zkey_profile.py > $R/gpurun_out/zkprof.log 2>&1 )
echo rc=$?; tail -3 gpurun_out/zkprof.log | cut -c1-400
f=$(find gpurun_out/zkprof -name "*kernel_stats.csv" | head -1); head -12 $f
