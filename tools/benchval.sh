#!/bin/bash
# usage: tools/benchval.sh <bench.py args...>  -> one short line (value, ms/step, kernel ms) + any "[glm debug]" line
python bench.py "$@" 2>/tmp/benchval.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('value %.4g %s  ms/step %.3f  kernel_ms %s  frac %.3f  parity %s  bound %s' % (d['value'], d['unit'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('frac') or 0, d.get('parity_max_rel_dev'), d.get('error_bound')))
"
grep "glm debug" /tmp/benchval.err | tail -1
