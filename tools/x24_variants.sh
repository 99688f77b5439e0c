#!/bin/bash
# on the GPU box: the x24 race reproducer (tools/dbg_ws2.py) against several builds of the library
cp tap-clip_amd/csrc/libtapclip.so /tmp/cur.so
export TAPCLIP_X24=1
for v in "$@"; do
  if [ $v = cur ]; then cp /tmp/cur.so tap-clip_amd/csrc/libtapclip.so; else cp tools/libtapclip_$v.so tap-clip_amd/csrc/libtapclip.so; fi
  echo "== $v"
  TAPCLIP_DEBUG_STOP=5 timeout -k 10 120 python tools/dbg_ws2.py 2>/dev/null >/dev/null
  TAPCLIP_DEBUG_STOP=4 timeout -k 10 120 python tools/dbg_ws2.py 2>/dev/null > /tmp/out_$v.txt
  echo "hits: $(wc -l < /tmp/out_$v.txt)"; head -3 /tmp/out_$v.txt
done
cp /tmp/cur.so tap-clip_amd/csrc/libtapclip.so
