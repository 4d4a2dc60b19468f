#!/bin/bash
# usage (GPU box): tools/gpu_abl_qf.sh <variant> [<variant> ...]  -- quadform ms/launch of the timing ablations (SEERHIP_ROUTE qf=<variant>) on the probe's rows
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== qf=$v $(SEERHIP_ROUTE=qf=$v V=${V:-262144} L=${L:-4} python $R/tools/gpu_probe_lmm.py 2>&1 | grep 'quadform ms')"
done
