cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
C=tap-clip_amd/csrc
cp $C/libtapclip.so /tmp/cur_keep.so
cp tools/libtapclip_fl.so $C/libtapclip.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_hf.py -m gpu -x -q -k "encode_image or block_vs or full_batch or image_embeddings" > gpurun_out/r4f/fl_parity.log 2>&1; echo "fl parity rc=$?"; tail -3 gpurun_out/r4f/fl_parity.log
cp /tmp/cur_keep.so $C/libtapclip.so
for i in 1 2; do
  echo "== cur"; GEMM_BENCH_ONLY=fc tools/gemm_bench 30 2>&1 | tail -4
  echo "== full lines"; GEMM_BENCH_ONLY=fc tools/gemm_bench_alt 30 2>&1 | tail -4
done | tee gpurun_out/r4f/gemm_bench.log
BENCH_ARGS="--steps 50 --no-configs4" bash tools/abrun.sh cur fl 2>&1 | tee gpurun_out/r4f/ab.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward" > gpurun_out/r4f/bwd.log 2>&1; echo "bwd rc=$?"; grep "gradient floor\|context-gradient floor\|passed\|failed" gpurun_out/r4f/bwd.log | tail -12
