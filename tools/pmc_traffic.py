#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py` to the HBM-side bytes per
launch of the dominant kernel family (the gemm256 launches), with the gfx950 correction the guide
prescribes: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  Both counters are in KiB.

usage: tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json>"""
import csv
import glob
import json
import re
import sys


def per_kernel(pass_dir, counter):
    f = glob.glob(f"{pass_dir}/**/*_counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
        short = m.group(1) if m else name[:60]
        e = acc.setdefault(short, [0.0, 0])
        e[0] += float(r["Counter_Value"])
        e[1] += 1
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "bytes per launch", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide reads); WRITE_SIZE KiB x 1024",
           "kernels": {}}
    tot_b = tot_n = 0
    for k in sorted(set(fetch) | set(write)):
        fb, fn = fetch.get(k, [0.0, 0])
        wb, wn = write.get(k, [0.0, 0])
        n = max(fn, wn)
        if n == 0:
            continue
        rd, wr = fb * 1024 * 2 / max(fn, 1), wb * 1024 / max(wn, 1)
        out["kernels"][k] = {"launches": n, "fetch_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
        if k.startswith("gemm256_kernel<0") or k.startswith("gemm256_kernel<1"):
            tot_b += (rd + wr) * n
            tot_n += n
    out["gemm_family_hbm_bytes_per_launch"] = round(tot_b / max(tot_n, 1))
    # which kernel sources these passes measured: bench.py reports `traffic` only while the library is built from them
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_sha16
    out["gemm_source_sha16"] = source_sha16()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
    for k, v in out["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
