#!/bin/bash
# Dev helper: one bench.py run per line of env assignments read from stdin, e.g.
#   printf 'RM_TILE_ORDER=natural\nRM_FORCE_FAST_FP=1 RM_FORCE_GENERIC_POW=1\n' | bash profiles/ab_env.sh [bench args]
# (separate processes, box-to-box and run-to-run noise is ~2 %: repeat the lines to
# interleave variants when the difference is small)
R=${GRAFT_REPO_ROOT:-$(pwd)}
while read -r line; do
  [ -z "$line" ] && continue
  env $line python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sizes --no-motion --check "$@" 2>/tmp/ab_err.log | LINE="$line" python3 -c "
import json,sys,os
t=sys.stdin.read()
line=os.environ['LINE'].replace(os.environ.get('GRAFT_REPO_ROOT','@@')+'/rusty-marcher_amd/lib/variants/','')
try:
    d=json.loads(t)
    print('%-58s kernel %7.1f us  step %7.1f us  %7.0f Mpx/s  maxdelta %.2e' % (line, d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value'], d['max_abs_delta_vs_oracle']))
except Exception as e:
    print(line, 'FAILED', t[:200], open('/tmp/ab_err.log').read()[-600:])"
done
