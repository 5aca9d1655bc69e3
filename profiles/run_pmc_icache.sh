#!/bin/bash
# Instruction- and scalar-cache counters of bench.py's render kernel (one --pmc pass per group).
# usage (GPU box, repo root): BENCH=... bash profiles/run_pmc_icache.sh <tag> [bench args]
set -u
TAG=${1:-ic}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH=${BENCH:-$R/bench.py}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $BENCH --steps 5 --warmup 2 --no-cpu-baseline --no-sizes "$@" > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rm_render_" not in row["Kernel_Name"]:
            continue
        a = agg[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    print("%s,%.1f,%d" % (k, agg[k][0] / agg[k][1], agg[k][1]))
PY
