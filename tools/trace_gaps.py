"""Idle time between consecutive kernels of the timed steps, from a rocprofv3 --kernel-trace run
(`rocprofv3 --kernel-trace -d DIR -o tr -- python3 bench.py ...` writes DIR/tr_results.db).
usage: python tools/trace_gaps.py DIR/tr_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, name from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "im2col8" in r[2]]  # a step starts at each im2col8 launch
print("kernels", len(rows), "steps", len(starts))
for a, b in zip(starts[-4:-1], starts[-3:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    print(f"kernels {len(seg)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {(span - busy) / 1e3:.1f} us  "
          f"median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us  max gap {max(gaps) / 1e3:.1f} us")
seg = rows[starts[-2]:starts[-1]]
for s, e, n in seg[:16]:
    print(f"{(s - seg[0][0]) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n[:80]}")
