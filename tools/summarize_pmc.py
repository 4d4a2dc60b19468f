#!/usr/bin/env python
"""Collapse rocprofv3 counter_collection / kernel_stats CSVs under gpurun_out/<round>/ into the small summaries kept in
profiles/<round>/ (per pass, kernel and counter: dispatches + mean per dispatch; traffic_lmm.json for bench.py)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main(src, dst, vpd):
    os.makedirs(dst, exist_ok=True)
    rows = []
    quad = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        per = collections.defaultdict(lambda: collections.defaultdict(float))     # a counter may be split over rows (XCDs)
        for r in csv.DictReader(open(files[0])):
            per[(short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])]["v"] += float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            agg[(k, c)].append(v["v"])
        for (k, c), v in sorted(agg.items()):
            rows.append((os.path.basename(d), k, c, len(v), sum(v) / len(v)))
            if "quadform" in k:
                quad[c] = sum(v) / len(v)
    with open(os.path.join(dst, "rocprofv3_pmc_summary_lmm_V%d.csv" % vpd), "w") as f:
        f.write("pass,kernel,counter,dispatches,mean_per_dispatch\n")
        for r in rows:
            f.write("%s,%s,%s,%d,%.6g\n" % r)
    if "FETCH_SIZE" in quad:
        json.dump({"kernel": "k_lmm_quadform_i8", "variants_per_dispatch": vpd, "FETCH_SIZE_KB": quad["FETCH_SIZE"],
                   "WRITE_SIZE_KB": quad.get("WRITE_SIZE"), "fetch_correction": 2.0,
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gpu_probe_lmm.py, "
                             "N=5000, L=5; gfx950: FETCH_SIZE counts 64 B per 128-B request, hence x2 "
                             "(MI355X_MICROARCH.md HBM section)"},
                  open(os.path.join(dst, "traffic_lmm.json"), "w"), indent=1)
    st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        with open(st[0]) as f, open(os.path.join(dst, "rocprofv3_kernel_stats_bench.csv"), "w") as g:
            g.write(f.read())
    for n in ("bench_n1.json",):
        p = os.path.join(src, n)
        if os.path.exists(p) and os.path.getsize(p):
            open(os.path.join(dst, n), "w").write(open(p).read())
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 262144)
