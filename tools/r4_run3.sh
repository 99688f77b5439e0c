cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm" > gpurun_out/r4d/gemm_tests.log 2>&1; echo "gemm tests rc=$?"; tail -5 gpurun_out/r4d/gemm_tests.log
python -m pytest tests/test_gpu_tied.py tests/test_gpu_hf.py -m gpu -x -q > gpurun_out/r4d/tied_hf.log 2>&1; echo "tied/hf rc=$?"; tail -5 gpurun_out/r4d/tied_hf.log
for lat in 1 0; do for p in bf16 fp16; do
  echo "== LAT=$lat $p" | tee -a gpurun_out/r4d/phases.log
  TAPCLIP_GEMM_LAT=$lat timeout -k 10 300 python tools/train_phases.py $p 2>&1 | grep -v "cls_specific\|amdgpu.ids" | tee -a gpurun_out/r4d/phases.log
done; done
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d/trace_$p -o t -- python3 $GRAFT_REPO_ROOT/tools/text_bench.py $p > $GRAFT_REPO_ROOT/gpurun_out/r4d/text_bench_$p.log 2>&1
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/r4d/trace_$p/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r4d/trace_$p/*/*kernel_stats.csv 2>/dev/null | head -n 1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r4d/text_kernel_stats_$p.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4d/trace_$p
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/r4d/text_bench_$p.log
  head -25 $GRAFT_REPO_ROOT/gpurun_out/r4d/text_kernel_stats_$p.csv | cut -c1-200
done
