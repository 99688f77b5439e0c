#!/bin/bash
# on the GPU box: the long-sequence attention kernels after the asm maxima were replaced (profiles/r05_flash2_asm_hazard.txt):
# checks at several score scales and sequence lengths (one / two query tiles per wave, with and without a tail block), both
# operand types, natural and base-2 scores; then the builds that KEEP packed-fp32 ops, whole outputs compared launch after launch
out=gpurun_out/$1; mkdir -p $out
log=$out/attn_hazard_check.log
for b in attn_bench attn_bench_f16; do for sc in 0.3 1.0 2.0; do for T in 577 576 528 513 257; do
  echo "== $b scale $sc T $T" >> $log
  ATTN_BENCH_SCALE=$sc ATTN_BENCH_T=$T timeout -k 10 100 tools/$b 4 1 1 0 10000 2>&1 | grep -v median >> $log
done; done; done
for b in attn_bench_pk attn_bench_f_pk; do
  echo "== $b (packed-fp32 ops kept), 128 sequences, 8 launches each" >> $log
  timeout -k 10 200 tools/$b 128 2 1 0 0 0 0 0 0 0 0 10000 10000 10000 10000 10000 10000 10000 10000 2>&1 | grep -v median >> $log
done
echo "== timing" >> $log
timeout -k 10 200 tools/attn_bench 128 16 1 0 10000 2>&1 | grep median >> $log
timeout -k 10 200 tools/attn_bench_f16 128 16 1 0 10000 2>&1 | grep median >> $log
timeout -k 10 200 tools/attn_bench_pk 128 16 1 0 10000 2>&1 | grep median >> $log
grep -c "CHECK OK" $log; grep -c "CHECK FAILED" $log; grep "FAILED" -B4 $log | head -40; tail -12 $log
