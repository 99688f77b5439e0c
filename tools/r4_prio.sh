cd $GRAFT_REPO_ROOT
for round in 1 2; do for pr in 0 -1; do for p in bf16 fp16; do
  echo "== prio=$pr $p"; TAPCLIP_IMAGE_STREAM_PRIORITY=$pr timeout -k 10 300 python tools/train_phases.py $p 2>&1 | grep "forward, no grad\|full step"
done; done; done | tee gpurun_out/r4_prio.log
