#!/bin/bash
# Dev helper like ab_env.sh, for the camera_on_the_move leg: one bench.py run per line of stdin, "CONFIG env assignments...", e.g.
#   printf 'C3 RM_FIRST_ROUND=6144\nC2 A=1\n' | bash profiles/ab_motion.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
while read -r cfg line; do
  [ -z "$cfg" ] && continue
  env $line python3 $R/bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-sizes "$@" 2>/tmp/ab_err.log | LINE="$cfg $line" python3 -c "
import json,sys,os
t=sys.stdin.read()
try:
    d=json.loads(t); m=d['camera_on_the_move']; f=m['first_frame_of_a_view']
    print('%-50s standing %6.1f  moving %6.1f  first %6.1f  fourth %6.1f us' % (os.environ['LINE'], d['roofline']['kernel_ms']*1e3, m['kernel_ms']*1e3, f['kernel_ms_median']*1e3, f['fourth_frame_of_the_view_kernel_ms_median']*1e3))
except Exception as e:
    print(os.environ['LINE'], 'FAILED', repr(e), t[:200], open('/tmp/ab_err.log').read()[-600:])"
done
