#!/usr/bin/env python3
"""Average duration of the image tower's GEMM launches (the roofline object's kernel family) from a
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py` run.  The per-symbol averages of
*_kernel_stats.csv mix these 256-workgroup launches with the text tower's and the backward's small grids of the
same kernel symbol; this filters the trace to the persistent full-chip launches (grid = 256 workgroups).

usage: tools/trace_family.py <dir with *_kernel_trace.csv> <out.json> [bf16|fp8]   (which precision's GEMM family; the
bench run also launches the other precisions' towers for its `precisions` table)"""
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm256_kernel<" not in n and "gemm_mx8_kernel<" not in n:
        continue
    if int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) != 256:
        continue
    short = n[n.index("gemm"):n.index(">") + 1]
    e = acc.setdefault(short, [0.0, 0])
    e[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e[1] += 1
which = sys.argv[3] if len(sys.argv) > 3 else "bf16"
if which == "fp8":
    fam = {k: v for k, v in acc.items() if k.startswith("gemm_mx8") and k.split("<")[1].rstrip(">") in ("0", "6")}
else:
    fam = {k: v for k, v in acc.items() if k.startswith("gemm256") and k.split("<")[1].split(",")[0] in ("0", "1") and ", false, 256" in k}
tot = sum(v[0] for v in fam.values())
n = sum(v[1] for v in fam.values())
out = {"family": which, "note": "full-chip (256-workgroup) launches only; QKV/out_proj/c_proj = epilogue 0, c_fc+GELU = epilogue 1 (bf16) / 6 (fp8)",
       "per_kernel": {k: {"launches": v[1], "avg_us": round(v[0] / v[1], 2)} for k, v in sorted(acc.items())},
       "gemm_family": {"launches": n, "avg_us": round(tot / n, 2)}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["gemm_family"]))
