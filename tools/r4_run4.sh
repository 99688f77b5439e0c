cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
run() { echo "== $1 :: $2" | tee -a gpurun_out/r4e/phases.log; env $2 timeout -k 10 300 python tools/train_phases.py $1 2>&1 | grep -v "cls_specific\|amdgpu.ids" | tee -a gpurun_out/r4e/phases.log; }
for round in 1 2; do
run bf16 "TAPCLIP_GEMM_LAT=0"
run bf16 "TAPCLIP_GEMM_LAT=1"
run bf16 "TAPCLIP_GEMM_LAT=1 TAPCLIP_GEMM_LAT_TILE=0"
run fp16 "TAPCLIP_GEMM_LAT=1"
run fp16 "TAPCLIP_GEMM_LAT=1 TAPCLIP_GEMM_LAT_TILE=0"
done
