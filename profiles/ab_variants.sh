#!/bin/bash
# Dev helper: run bench.py once per library variant in rusty-marcher_amd/lib/variants/
# (RM_LIB_PATH override) and print kernel time per variant.  usage: bash profiles/ab_variants.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for so in $R/rusty-marcher_amd/lib/variants/*.so; do
  RM_LIB_PATH=$so python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --check "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s kernel %.1f us  step %.1f us  %.0f Mpx/s  maxdelta %.2e' % (d['config']['build'].split()[1], d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3, d['value'], d['max_abs_delta_vs_oracle']))"
done
