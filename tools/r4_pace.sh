cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fullmodel_tiny" > gpurun_out/r4_pace_sanity.log 2>&1; echo "sanity rc=$?"
TAPCLIP_PACE_K=5 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "fullmodel" > gpurun_out/r4_pace_tests.log 2>&1; echo "paced tests rc=$?"; tail -2 gpurun_out/r4_pace_tests.log
for round in 1 2; do for k in 0 3 5 8 12; do for p in bf16 fp16; do
  echo "== pace=$k $p"; TAPCLIP_PACE_K=$k timeout -k 10 300 python tools/train_phases.py $p 2>&1 | grep "forward, no grad\|full step"
done; done; done | tee gpurun_out/r4_pace.log
