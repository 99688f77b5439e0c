#!/usr/bin/env python3
"""Reduce one `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -- python3 bench.py ...` pass to
per-kernel MFMA utilisation: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
(GRBM_GUI_ACTIVE is summed over the 8 XCDs), clock = GRBM_GUI_ACTIVE / 8 / duration.

usage: tools/pmc_sq.py <pass dir> <out.json>"""
import csv
import glob
import json
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
disp = {}
for r in csv.DictReader(open(f)):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
acc = {}
for d in disp.values():
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", d["name"])
    short = m.group(1) if m else d["name"][:60]
    a = acc.setdefault(short, {"n": 0, "dur": 0.0, "gui": 0.0, "mfma": 0.0, "wave": 0.0, "wait": 0.0, "wait_inst": 0.0, "active": 0.0, "ldsbc": 0.0})
    a["n"] += 1
    a["dur"] += d["dur"]
    a["gui"] += d.get("GRBM_GUI_ACTIVE", 0)
    a["mfma"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    a["wave"] += d.get("SQ_WAVE_CYCLES", 0)
    a["wait"] += d.get("SQ_WAIT_ANY", 0)
    a["wait_inst"] += d.get("SQ_WAIT_INST_ANY", 0)
    a["active"] += d.get("SQ_ACTIVE_INST_ANY", 0)
    a["ldsbc"] += d.get("SQ_LDS_BANK_CONFLICT", 0)
out = {"note": "rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE over bench.py --steps 3; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); clock = GRBM_GUI_ACTIVE/8/duration",
       "kernels": {}}
for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["dur"]):
    if a["gui"] == 0 or a["wave"] == 0:
        continue
    cyc = a["gui"] / 8
    out["kernels"][k] = {"launches": a["n"], "avg_us": round(a["dur"] / a["n"] / 1e3, 1), "clock_GHz": round(cyc / a["dur"], 2),
                         "mfma_util": round(a["mfma"] / (1024 * cyc), 3), "wait_any_frac": round(a["wait"] / a["wave"], 3),
                         "wait_inst_frac": round(a["wait_inst"] / a["wave"], 3), "active_frac": round(a["active"] / a["wave"], 3),
                         "lds_bank_conflict_cycles_per_launch": int(a["ldsbc"] / a["n"])}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in list(out["kernels"].items())[:8]:
    print(k, v)
