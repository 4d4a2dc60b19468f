#!/usr/bin/env python
"""Fill the R02_* placeholders of profiles/README.md from profiles/r02/*.json and the kernel-stats CSVs (run after tools/summarize_r02.py)."""
import csv, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(R, "profiles", "r02")
def line(cfg):
    return json.loads(open(os.path.join(D, "bench_%s.json" % cfg)).read().strip().splitlines()[-1])
def prof_ms(cfg, key):
    for r in csv.DictReader(open(os.path.join(D, "rocprofv3_kernel_stats_%s.csv" % cfg))):
        if r["Name"].startswith(key):
            return float(r["AverageNs"]) / 1e6
    return float("nan")
c3 = line("C3"); sub = {}
sub["R02_C3_VALUE"] = "%.1f" % (c3["value"] / 1e6); sub["R02_C3_MS"] = "%.1f" % c3["ms_per_step"]
sub["R02_C3_KMS"] = "%.2f" % c3["roofline"]["kernel_ms"]; sub["R02_C3_PROF"] = "%.2f" % prof_ms("C3", "k_lmm_quadform_i8<0>")
sub["R02_C3_TOPS"] = "%.0f" % c3["roofline"]["achieved"]; sub["R02_C3_FRAC"] = "%.3f" % c3["roofline"]["frac"]
sub["R02_C3_TRAFFIC"] = "%.1f" % ((c3["roofline"]["traffic"] or 0) / 1e9)
cb = c3.get("cpu_baseline", {}); sub["R02_C3_CPU"] = "%.0f" % cb.get("value", float("nan")); sub["R02_C3_CORES"] = str(cb.get("cores", "?"))
sub["R02_C3_PERCORE"] = "%.0f/s" % (cb.get("value", float("nan")) / max(cb.get("cores", 1), 1))
for cfg, tag in (("C2", "C2"), ("C2N5000", "C2N"), ("C4", "C4")):
    d = line(cfg); rf = d["roofline"]
    sub["R02_%s_VALUE" % tag] = "%.2f" % (d["value"] / 1e6); sub["R02_%s_KMS" % tag] = "%.1f" % rf["kernel_ms"]
    sub["R02_%s_F64" % tag] = "%.1f" % (rf["achieved"] or float("nan")); sub["R02_%s_FRAC" % tag] = "%.0f %%" % (100 * (rf["frac"] or float("nan")))
    f32 = (rf.get("fp32_flops_per_variant") or 0.0) * d["config"]["variants_per_step_per_gpu"] / (rf["kernel_ms"] * 1e-3) / 1e12
    sub["R02_%s_F32" % tag] = "%.1f" % f32
    sub["R02_%s_CPU" % tag] = "%.0f" % d.get("cpu_baseline", {}).get("value", float("nan"))
p = os.path.join(R, "profiles", "README.md")
s = open(p).read()
for k in sorted(sub, key=len, reverse=True):
    s = s.replace(k, sub[k])
open(p, "w").write(s)
print(sub)
