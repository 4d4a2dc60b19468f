#!/usr/bin/env python
"""Collapse the rocprofv3 outputs of tools/profile_r05.sh (gpurun_out/prof_r05/; round 4: profile_r04.sh) into the small files kept under profiles/rNN/:
  bench_<cfg>.json                         the bench lines
  rocprofv3_kernel_stats_<cfg>.csv         --kernel-trace --stats of the same command
  rocprofv3_pmc_summary_<cfg>.csv          per pass / kernel / counter: dispatches, sum, mean per dispatch
  flops_<cfg>.json, traffic_<cfg>.json     what bench.py scales to its launch size for `roofline`
"""
import collections
import csv
import glob
import json
import os
import re
import sys

VARIANTS = {"C3": 1 << 20, "C2": 1 << 20, "C2N5000": 1 << 18, "C4": 1 << 18}      # variants per step of bench.py --config <cfg> (the PMC runs time ONE step)
GLM = ("k_glm_", "k_firth_")


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def read_pass(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = collections.defaultdict(float)
    for f in files:
        for r in csv.DictReader(open(f)):
            per[(short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for (k, c, _), v in per.items():
        agg[(k, c)].append(v)
    return agg


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for cfg, V in VARIANTS.items():
        rows = []
        tot = collections.defaultdict(float)           # counter -> sum over the dominant kernels of one step
        for d in sorted(glob.glob(os.path.join(src, "pmc_%s_*" % cfg))):
            if not os.path.isdir(d):
                continue
            for (k, c), v in sorted(read_pass(d).items()):
                if not k.startswith("k_"):                    # torch's data-generation kernels are not the subject
                    continue
                rows.append((os.path.basename(d), k, c, len(v), sum(v), sum(v) / len(v)))
                dominant = ("quadform_i8<0>" in k or "quadform_i8w<0, 0, 0>" in k or k.endswith("k_lmm_quadform_i8") or k.endswith("k_lmm_quadform_i8w")) if cfg == "C3" else any(g in k for g in GLM)
                if dominant:
                    tot[c] += sum(v)
        if rows:                                              # a run with PMC=0 has no counter passes: bench line and kernel stats only
            with open(os.path.join(dst, "rocprofv3_pmc_summary_%s.csv" % cfg), "w") as f:
                f.write("pass,kernel,counter,dispatches,sum,mean_per_dispatch\n")
                for r in rows:                                # (kernel names hold commas: quoted since round 5)
                    f.write('%s,"%s",%s,%d,%.6g,%.6g\n' % r)
            # where the bytes go, per kernel (round-4 review item 6): FETCH_SIZE x 2 (gfx950) and WRITE_SIZE of one step
            per = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for _, k, c, n, s_, _m in rows:
                if c == "FETCH_SIZE": per[k][0] = n; per[k][1] = s_ * 2.0 * 1024.0
                if c == "WRITE_SIZE": per[k][0] = n; per[k][2] = s_ * 1024.0
            if per:
                with open(os.path.join(dst, "traffic_per_kernel_%s.csv" % cfg), "w") as f:
                    f.write("kernel,dispatches,fetch_bytes,write_bytes,bytes_per_variant\n")
                    for k, (n, fb, wb) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
                        f.write('"%s",%d,%.0f,%.0f,%.1f\n' % (k, n, fb, wb, (fb + wb) / V))
        tag = "lmm" if cfg == "C3" else cfg.lower()
        if "FETCH_SIZE" in tot:
            json.dump({"kernels": "k_lmm_quadform_i8w" if cfg == "C3" else "all k_glm_* / k_firth_* kernels of one step",
                       "variants_per_dispatch": V, "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot.get("WRITE_SIZE"),
                       "fetch_correction": 2.0,
                       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_r05.sh) on %s; gfx950: FETCH_SIZE "
                                 "counts 64 B per 128-B request, hence x2 (MI355X_MICROARCH.md HBM section)"
                                 % ("bench.py --config %s --steps 1 --warmup 0 (the bench's own rows)" % cfg)},
                      open(os.path.join(dst, "traffic_%s.json" % tag), "w"), indent=1)
        if "SQ_INSTS_VALU_FMA_F64" in tot:
            f64 = 64.0 * (2 * tot["SQ_INSTS_VALU_FMA_F64"] + tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + tot["SQ_INSTS_VALU_TRANS_F64"])
            f32 = 64.0 * (2 * tot.get("SQ_INSTS_VALU_FMA_F32", 0) + tot.get("SQ_INSTS_VALU_ADD_F32", 0) + tot.get("SQ_INSTS_VALU_MUL_F32", 0)
                          + tot.get("SQ_INSTS_VALU_TRANS_F32", 0)) + 512.0 * tot.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)
            f16 = 512.0 * tot.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0)
            json.dump({"variants_per_step": V, "fp64_flops_per_variant": f64 / V, "fp32_flops_per_variant": f32 / V, "fp16_mfma_flops_per_variant": f16 / V,
                       "counters": {k: v for k, v in tot.items() if k.startswith("SQ_INSTS_VALU")},
                       "source": "rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 / _F32 + MFMA_MOPS_F32 (wave-level instruction counts x 64 lanes; "
                                 "FMA = 2 flop; MFMA_MOPS_F32 in units of 512 flop) summed over every k_glm_* / k_firth_* dispatch of "
                                 "`bench.py --config %s --steps 1 --warmup 0`" % cfg},
                      open(os.path.join(dst, "flops_%s.json" % tag), "w"), indent=1)
        for sdir, suffix in (("stats_%s" % cfg, ""), ("stats1_%s" % cfg, "_one_lane")):
            st = glob.glob(os.path.join(src, sdir, "**", "*kernel_stats.csv"), recursive=True)
            if st:                                                # this library's kernels in full; torch's (data generation, eigh) as one line
                rows_in = list(csv.DictReader(open(st[0])))
                with open(os.path.join(dst, "rocprofv3_kernel_stats_%s%s.csv" % (cfg, suffix)), "w") as f:
                    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
                    oc, ot = 0, 0.0
                    for r in rows_in:
                        nm = short(r["Name"])
                        if nm.startswith("k_") or nm.startswith("__amd_rocclr"):
                            f.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (nm, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))
                        else:
                            oc += int(r["Calls"]); ot += float(r["TotalDurationNs"])
                    f.write('"(torch / rocSOLVER kernels: synthetic data generation, eigendecomposition)",%d,%.0f,,,,,\n' % (oc, ot))
        b = os.path.join(src, "bench_%s.json" % cfg)
        if os.path.exists(b) and os.path.getsize(b):
            open(os.path.join(dst, "bench_%s.json" % cfg), "w").write(open(b).read())
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
