#!/bin/bash
# on the GPU box: interleaved pipeline A/B of library variants (tools/libtapclip_<v>.so; "cur" = the in-tree library)
cp tap-clip_amd/csrc/libtapclip.so /tmp/cur.so
trap 'cp /tmp/cur.so tap-clip_amd/csrc/libtapclip.so' EXIT  # an interrupted run must not leave an A/B build installed
for round in 1 2; do for v in "$@"; do
  if [ $v = cur ]; then cp /tmp/cur.so tap-clip_amd/csrc/libtapclip.so; else cp tools/libtapclip_$v.so tap-clip_amd/csrc/libtapclip.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-input-side --no-precisions --no-full-forward $BENCH_ARGS 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); k=r['kernels']; print(r['value'], r['ms_per_step'], {n:k[n]['avg_us'] for n in k})"
done; done
cp /tmp/cur.so tap-clip_amd/csrc/libtapclip.so
