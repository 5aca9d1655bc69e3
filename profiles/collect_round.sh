#!/bin/bash
# Collects what profiles/ keeps for a round, on the GPU box (from the repo root):
#   bash profiles/collect_round.sh r03
# (1) rocprofv3 --kernel-trace --stats of bench.py for C2 (the headline command), C3, C5, C4;
# (2) PMC counters of the same launches, one --pmc pass per group (profiles/run_pmc.sh), all four configs;
# (3) the default bench line of the same build, and `rocprofv3 --kernel-trace --stats -- python3 bench.py` itself.
# Everything lands under gpurun_out/<tag>_*; profiles/summarise_round.py turns it into the
# committed summaries and profiles/pmc_<config>.json.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in C2 C3 C5 C4; do
  steps=200; [ $c = C5 ] && steps=20; [ $c = C4 ] && steps=20
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$c -- python3 $R/bench.py --config $c --steps $steps --warmup 10 --no-cpu-baseline --no-sizes --no-motion > $R/gpurun_out/${TAG}_trace_$c.json 2> $R/gpurun_out/${TAG}_trace_$c.err || echo "trace $c failed"
  find $R/gpurun_out/${TAG}_trace_$c -name "*kernel_trace.csv" -delete      # (one row per launch: megabytes; the stats file is what is kept)
  echo "trace $c done"
done
cd $R
for c in C2 C3 C5 C4; do
  bash profiles/run_pmc.sh ${TAG}_$c --config $c > /dev/null 2>&1 || echo "pmc $c failed"
  echo "pmc $c done"
done
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench done"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> gpurun_out/${TAG}_bench_driver_cmd.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_default -- python3 $R/bench.py > $R/gpurun_out/${TAG}_trace_default.json 2> $R/gpurun_out/${TAG}_trace_default.err || echo "trace default failed"
find $R/gpurun_out/${TAG}_trace_default -name "*kernel_trace.csv" -delete
cd $R
ls gpurun_out | grep ${TAG}_ | head -60
