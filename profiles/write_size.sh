#!/bin/bash
# Dev helper: WRITE_SIZE / FETCH_SIZE (KB per launch) of the render kernel for the current env (RM_LIB_PATH etc).
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d /tmp/ws.XXXX)
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/$c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sizes "$@" > /dev/null 2> $T/err.log
python3 - $T/$c $c <<'PY'
import csv, glob, sys
v=[float(r["Counter_Value"]) for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "rm_render_" in r["Kernel_Name"]]
print("%s mean per launch: %.1f (n=%d)" % (sys.argv[2], sum(v)/max(len(v),1), len(v)))
PY
done
