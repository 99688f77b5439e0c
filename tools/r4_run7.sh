cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
TAPCLIP_GN="5,3,6,3" python -m pytest tests/test_gpu_parity.py tests/test_gpu_hf.py -m gpu -x -q -k "encode_image or block_vs or full_batch or image_embeddings" > gpurun_out/r4h/gn_parity.log 2>&1; echo "gn parity rc=$?"; tail -2 gpurun_out/r4h/gn_parity.log
for round in 1 2; do for GN in "0,0,0,0" "5,3,6,3" "0,3,6,0" "0,0,8,0" "0,0,4,0" "5,3,6,0"; do
  echo "== GN=$GN"
  TAPCLIP_GN=$GN timeout -k 10 200 python bench.py --steps 50 --no-cpu-baseline --no-input-side --no-precisions --no-full-forward --no-configs4 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); k=r['kernels']; print(r['value'], r['ms_per_step'], {n:k[n]['avg_us'] for n in k if n.startswith('gemm')})"
done; done | tee gpurun_out/r4h/gn_ab.log
