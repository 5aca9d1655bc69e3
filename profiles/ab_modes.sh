#!/bin/bash
# Dev helper: bench.py once per RM_KERNEL_MODE value.  usage: bash profiles/ab_modes.sh "s4x4 s1x1 p4c4" [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
MODES=$1; shift
for m in $MODES; do
  RM_KERNEL_MODE=$m python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --check "$@" 2>/tmp/ab_err.log | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    print('%-8s kernel %.1f us  step %.1f us  %.0f Mpx/s  maxdelta %.2e' % ('$m', d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value'], d['max_abs_delta_vs_oracle']))
except Exception as e:
    print('$m', 'FAILED', t[:200], open('/tmp/ab_err.log').read()[-400:])"
done
