#!/bin/bash
# on the GPU box: the four image-tower GEMM shapes at the row counts of small batches, under the launcher's knobs
out=gpurun_out/$1; mkdir -p $out
for M in 1576 6304 12608 25216; do
  for cfg in "default" "TAPCLIP_GEMM_BN=256" "TAPCLIP_GEMM_BN=128" "TAPCLIP_GEMM_BN=256 TAPCLIP_NO_TAIL_SPLIT=1" "TAPCLIP_GEMM_BN=256 TAPCLIP_TAIL_MIN_KS=8" "TAPCLIP_GEMM256_MIN_M=100000" "TAPCLIP_GEMM256_MIN_M=100000 TAPCLIP_GEMM_LAT_TILE=0" "TAPCLIP_GEMM256_MIN_M=100000 TAPCLIP_GEMM_LAT_TILE=1" "TAPCLIP_GEMM256_MIN_M=100000 TAPCLIP_GEMM_LAT_TILE=2"; do
    echo "== M=$M $cfg" >> $out/gemm_sweep.log
    if [ "$cfg" = default ]; then timeout -k 5 60 tools/gemm_bench 10 $M image N >> $out/gemm_sweep.log 2>&1; else env $cfg timeout -k 5 60 tools/gemm_bench 10 $M image N >> $out/gemm_sweep.log 2>&1; fi
  done
done
grep -c median $out/gemm_sweep.log
