#!/bin/bash
# on the GPU box: tools/train_phases.py (image tower alone, text features, overlapped forward, training step) with the shipped
# libraries ("nopk": six more kernel files without packed-fp32 ops) against the A/B build that keeps them ("pk"), interleaved
C=tap-clip_amd/csrc
cp $C/libtapclip.so /tmp/nopk.so; cp $C/libtapclip_fp16.so /tmp/nopk16.so
# whatever ends this script (a timeout, a lease end, Ctrl-C) the SHIPPED libraries go back
trap 'cp /tmp/nopk.so $C/libtapclip.so; cp /tmp/nopk16.so $C/libtapclip_fp16.so' EXIT
for round in 1 2; do for v in nopk pk; do
  if [ $v = nopk ]; then cp /tmp/nopk.so $C/libtapclip.so; cp /tmp/nopk16.so $C/libtapclip_fp16.so
  else cp tools/libtapclip_pk.so $C/libtapclip.so; cp tools/libtapclip_fp16_pk.so $C/libtapclip_fp16.so; fi
  for p in bf16 fp16; do echo "== $v $p"; timeout -k 10 300 python tools/train_phases.py $p 2>&1 | grep -v "cls_specific\|amdgpu.ids"; done
done; done
cp /tmp/nopk.so $C/libtapclip.so; cp /tmp/nopk16.so $C/libtapclip_fp16.so
