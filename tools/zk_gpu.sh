cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_zkey.py -x -q --timeout=900 --durations=10 > gpurun_out/zkey_tests.log 2>&1
echo "rc=$?"; tail -30 gpurun_out/zkey_tests.log
