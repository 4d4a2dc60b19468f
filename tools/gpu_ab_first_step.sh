#!/bin/bash
# A/B of the bordered first Newton step (k_glm_first_step) against the general kernel (SEERHIP_ROUTE=first_bordered=0), one gpurun call
for cfg in C2 C2N5000; do
  for v in 0 1 0 1; do
    echo -n "$cfg bordered=$v: "
    SEERHIP_ROUTE=first_bordered=$v python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g variants/s  %.3f ms/step  parity %s' % (r['value'], r['ms_per_step'], r.get('parity_max_rel_dev')))"
  done
done
