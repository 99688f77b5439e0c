set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_tied.py -x -q > gpurun_out/r4a/tied.log 2>&1; echo "tied rc=$?" >> gpurun_out/r4a/tied.log
tail -5 gpurun_out/r4a/tied.log
for p in bf16 fp16; do
  for tie in 1 0; do
    echo "== $p tie=$tie" >> gpurun_out/r4a/phases.log
    TAPCLIP_TIE_PADDING=$tie python tools/train_phases.py $p 2>&1 | grep -v cls_specific >> gpurun_out/r4a/phases.log
  done
done
cat gpurun_out/r4a/phases.log
for mm in 2048 1024; do
  echo "== fp16 tie=1 MIN_M=$mm" >> gpurun_out/r4a/phases.log
  TAPCLIP_GEMM256_MIN_M=$mm python tools/train_phases.py fp16 2>&1 | grep -v cls_specific >> gpurun_out/r4a/phases.log
  echo "== bf16 tie=1 MIN_M=$mm" >> gpurun_out/r4a/phases.log
  TAPCLIP_GEMM256_MIN_M=$mm python tools/train_phases.py bf16 2>&1 | grep -v cls_specific >> gpurun_out/r4a/phases.log
done
tail -40 gpurun_out/r4a/phases.log
