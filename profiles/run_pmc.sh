#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py's launches, one --pmc pass per counter group
# (separate runs, kernel-trace only -- never combined with sys traces).
# usage (on the GPU box, from the repo root):  bash profiles/run_pmc.sh <tag> [bench args]
# -> gpurun_out/pmc_<tag>/summary.csv: mean counter value per launch, for the render kernel
#    (rm_render_static) and for the classification kernel in front of it (rm_classify_tiles_kernel)
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH=${BENCH:-$R/bench.py}     # BENCH=build/old/bench.py profiles the bench of another tree
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
  "WRITE_SIZE" \
  "FETCH_SIZE" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  # (a short warm-up: every launch is a few hundred rows of counters, and what comes back from the box is bounded)
  RM_BENCH_WARMUP_S=0.02 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $BENCH --steps 5 --warmup 2 --no-cpu-baseline --no-sizes --no-motion "$@" > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        which = "render" if "rm_render_" in name else "classify" if "rm_classify_" in name else None
        if which is None:
            continue
        a = agg[(which, row["Counter_Name"])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/summary.csv", "w") as g:
    g.write("kernel,counter,mean_per_launch,launches\n")
    for k in sorted(agg):
        g.write("%s,%s,%.1f,%d\n" % (k[0], k[1], agg[k][0] / agg[k][1], agg[k][1]))
print(open(out + "/summary.csv").read())
PY
rm -rf $OUT/g[0-9]     # (the raw per-launch rows stay on the box: the summary and the bench lines are what is kept)
