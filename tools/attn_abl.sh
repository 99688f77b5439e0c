#!/bin/bash
# on the GPU box: the variant builds of tools/attn_bench one after the other (ablations are timing-only: their check fails)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  b=tools/attn_bench; [ "$v" != base ] && b=tools/attn_bench_$v
  echo "== $v" >> $out/attn_abl.log
  timeout -k 10 120 $b 128 8 $CFGS >> $out/attn_abl.log 2>&1 || true
done
grep "==\|median\|check" $out/attn_abl.log
