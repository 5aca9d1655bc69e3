#!/bin/bash
# Collects what profiles/ keeps for a round, on the GPU box (from the repo root):
#   bash profiles/collect_round.sh r02
# (1) rocprofv3 --kernel-trace --stats of bench.py for C2 (the default command), C3, C5, C4;
# (2) PMC counters of the same kernels, one --pmc pass per group (profiles/run_pmc.sh).
# Everything lands under gpurun_out/<tag>_*; profiles/summarise_round.py turns it into the
# committed summaries and profiles/pmc_latest.json.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in C2 C3 C5 C4; do
  steps=200; [ $c = C5 ] && steps=20; [ $c = C4 ] && steps=20
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$c -- python3 $R/bench.py --config $c --steps $steps --warmup 10 --no-cpu-baseline --no-sizes > $R/gpurun_out/${TAG}_trace_$c.json 2> $R/gpurun_out/${TAG}_trace_$c.err || echo "trace $c failed"
done
cd $R
for c in C2 C3 C5; do
  bash profiles/run_pmc.sh ${TAG}_$c --config $c > /dev/null 2>&1 || echo "pmc $c failed"
done
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
ls gpurun_out | grep ${TAG}_ | head -40
