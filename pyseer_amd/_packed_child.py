"""One device's share of a `--gpus` job over a packed cache, as a PROCESS of its own (round 6): `python -m pyseer_amd._packed_child <state dir> <i>`.

The reference's `--cpu N` is a pool of worker processes over blocks of the variant stream (pyseer/__main__.py:541-568, 777-780).  Round 5 ran the
streams of several devices as threads of the command line's process; measured with eight contexts on one device, eight threads of one
address space cost 0.066 CPU-s per million rows (page faults and hipHostRegister queue on one lock, the HIP runtime's helper threads
multiply), eight processes 0.022 (profiles/r06/host_budget.json).  So the parent does the run's host set-up once (phenotypes, null model or
kinship decomposition, lineage design), leaves it in a directory, and every device's process sets its engine up from there and runs the
library's own block loop (sh_job_run_packed) over its range of the cache: nothing is shared between the processes but the page cache.
Writes <state dir>/result_<i>.json: the counters and the CPU seconds of its block loop."""
import json
import os
import pickle
import resource
import sys
import time

import numpy as np


def main():
    d, i = sys.argv[1], int(sys.argv[2])
    st = pickle.load(open(os.path.join(d, "state.pkl"), "rb"))
    a = np.load(os.path.join(d, "arrays.npz"))
    from . import _abi, _route
    from .engine import Engine, Job
    lib = _abi.load()
    lib.sh_set_wait_mode(0 if _route.route("wait") == "spin" else 1)
    lib.sh_set_host_streams(len(st["devices"]))                # this process is one of that many: its share of the CPU budget and of the pinning budget
    dev = st["devices"][i]
    e = Engine(st["n"], device=dev)
    if st["lmm"]:
        e.lmm_setup(a["U"], a["S"], a["Y"], a["X"], float(st["h2"]), st["continuous"], st["filter_pvalue"], st["lrt_pvalue"])
    else:
        e.glm_setup(a["y"], a["W"] if a["W"].size else None, st["continuous"], st["llf"], st["firth_null"], st["filter_pvalue"], st["lrt_pvalue"])
    e.set_dedup(not st["no_dedup"])
    if st["lineage_labels"] is not None:
        e.lineage_setup(a["lin"], a["lin_cov"] if a["lin_cov"].size else None)
    e.set_af_filter(st["min_af"], st["max_af"])
    job = Job(e, st["lmm"], st["print_filtered"], lineage_labels=st["lineage_labels"], lineage_per_variant=st["lineage_per_variant"],
              patterns=st["patterns"], sample_names=st["sample_names"])
    out_fd = 1 if i == 0 else os.open(os.path.join(d, "out_%d.tsv" % i), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    pat_fd = -1
    if st["patterns"]:
        pat_fd = os.open(os.path.join(d, "pat_%d.txt" % i), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    ru0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter(); cpu0 = _abi.host_cpu_seconds()
    try:
        pf, te, pr, nb = job.run_packed(st["path"], (i, len(st["devices"])), st["job_block"], use_dma=st["dma"], out_fd=out_fd, pat_fd=pat_fd)
    finally:
        job.close()
    ru1 = resource.getrusage(resource.RUSAGE_SELF); cpu1 = _abi.host_cpu_seconds()
    e.close()
    res = {"prefilter": pf, "tested": te, "printed": pr, "blocks": nb, "wall_s": time.perf_counter() - t0,
           "user_s": ru1.ru_utime - ru0.ru_utime, "sys_s": ru1.ru_stime - ru0.ru_stime,
           "library_stage_cpu_s": {k: cpu1[k] - cpu0.get(k, 0.0) for k in cpu1}}
    with open(os.path.join(d, "result_%d.json" % i), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    try:
        main()
    except Exception as ex:                                   # the parent prints this stream's message and stops the others
        sys.stderr.write("pyseer_amd: device stream failed: %s\n" % ex)
        sys.exit(1)
