#!/bin/bash
# Dev helper: same-box A/B of two source trees (e.g. build/old = `git archive` of an earlier
# commit, built in place, against the working tree): alternates bench.py runs of each.
#   bash profiles/ab_trees.sh build/old . 3 --config C2 --steps 100
A=$1; B=$2; N=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $N); do
  for T in $A $B; do
    python3 $R/$T/bench.py --no-cpu-baseline --no-sizes "$@" 2>/tmp/abt_err.log | T=$T python3 -c "
import json,sys,os
t=sys.stdin.read()
try:
    d=json.loads(t); print('%-12s kernel %8.1f us   %8.0f Mpx/s   %s' % (os.environ['T'], d['roofline']['kernel_ms']*1e3, d['value'], d['roofline']['kernel']))
except Exception as e:
    print(os.environ['T'], 'FAILED', t[:200], open('/tmp/abt_err.log').read()[-500:])"
  done
done
