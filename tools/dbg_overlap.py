"""Analyse a rocprofv3 --kernel-trace CSV: do consecutive kernels of ONE queue overlap in time?"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Dispatch_Id")))
for q, ks in byq.items():
    ks.sort()
    n_ov = 0
    worst = 0
    ex = []
    for (s0, e0, n0, d0), (s1, e1, n1, d1) in zip(ks, ks[1:]):
        if s1 < e0:
            n_ov += 1
            if e0 - s1 > worst: worst = e0 - s1
            if len(ex) < 6: ex.append((n0, n1, e0 - s1))
    print(f"queue {q}: {len(ks)} kernels, {n_ov} overlapping successor pairs, worst overlap {worst} ns")
    for e in ex: print("    ", e)
