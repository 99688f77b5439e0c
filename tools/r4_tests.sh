cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -q > gpurun_out/r4c/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a gpurun_out/r4c/gpu_tests.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r4c/gpu_tests.log | tail -30
