#!/bin/bash
# Wave stamps of the round's kernels (RM_EXP_STAMPS build: make -C rusty-marcher_amd/csrc variant NAME=stamps DEFS=-DRM_EXP_STAMPS),
# on the GPU box from the repo root:  bash profiles/stamps_round.sh r02   -> gpurun_out/<tag>_stamps_<C>.txt
# (the frame stamped is the last of a few: with the feedback on it is dispatched by its predecessor's tile times)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$R/rusty-marcher_amd/lib/variants
for c in C2 C3 C5; do
  RM_LIB_PATH=$V/stamps/librusty_marcher_amd.so RM_DEBUG_STAMPS=$R/gpurun_out/st_$c.bin python3 $R/bench.py --config $c --steps 3 --warmup 3 --no-cpu-baseline --no-sizes > /dev/null 2> $R/gpurun_out/st_$c.err
  python3 $R/profiles/analyze_stamps.py $R/gpurun_out/st_$c.bin > $R/gpurun_out/${TAG}_stamps_$c.txt
  rm -f $R/gpurun_out/st_$c.bin
done
RM_FEEDBACK=0 RM_LIB_PATH=$V/stamps/librusty_marcher_amd.so RM_DEBUG_STAMPS=$R/gpurun_out/st_C5.bin python3 $R/bench.py --config C5 --steps 3 --warmup 3 --no-cpu-baseline --no-sizes > /dev/null 2> $R/gpurun_out/st_C5.err
python3 $R/profiles/analyze_stamps.py $R/gpurun_out/st_C5.bin > $R/gpurun_out/${TAG}_stamps_C5_no_feedback.txt
rm -f $R/gpurun_out/st_C5.bin
