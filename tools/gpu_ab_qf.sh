#!/bin/bash
# usage (GPU box): tools/gpu_ab_qf.sh <name> [<name> ...]  -- quadform ms/launch of A/B builds pyseer_amd/ab/libseerhip_<name>.so ("cur" = in-tree) on the probe's rows
R=$GRAFT_REPO_ROOT
for n in "$@"; do
  lib=$R/pyseer_amd/ab/libseerhip_$n.so; [ "$n" = "cur" ] && lib=$R/pyseer_amd/libseerhip.so
  echo "== $n $(SEERHIP_LIB=$lib SEERHIP_ROUTE=qf=${QF:-4} V=${V:-262144} L=${L:-4} python $R/tools/gpu_probe_lmm.py 2>&1 | grep 'quadform ms')"
done
