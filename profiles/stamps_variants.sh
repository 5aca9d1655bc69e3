#!/bin/bash
# Where a tile wave's time goes (diagnostic builds of the kernels: per-wave stamps, RM_DEBUG_STAMPS=<file> keeps the last launch's).
#   make -C rusty-marcher_amd/csrc variant NAME=stamps    DEFS=-DRM_EXP_STAMPS                        cull steps / candidates / bundles  -> stamps_cull.py
#   make -C rusty-marcher_amd/csrc variant NAME=walkstats DEFS="-DRM_EXP_STAMPS -DRM_EXP_WALKSTATS"   hierarchy walks, nodes, leaf tests -> stamps_walk.py
#   make -C rusty-marcher_amd/csrc variant NAME=phases    DEFS="-DRM_EXP_STAMPS -DRM_EXP_PHASES"      time in closest_hit / shadow walks / shading -> stamps_phases.py
#   make -C rusty-marcher_amd/csrc variant NAME=startup   DEFS="-DRM_EXP_STAMPS -DRM_EXP_STARTUP"     a wave's first microseconds, and its last -> stamps_startup.py
# usage (on the GPU box): bash profiles/stamps_variants.sh <variant> <script> [configs...]
R=${GRAFT_REPO_ROOT:-$PWD}
V=$1; S=$2; shift 2
mkdir -p $R/gpurun_out
for c in ${@:-C2 C3 C5}; do
  RM_LIB_PATH=$R/rusty-marcher_amd/lib/variants/$V/librusty_marcher_amd.so RM_DEBUG_STAMPS=$R/gpurun_out/sv.bin python3 $R/bench.py --config $c --steps 4 --warmup 4 --no-cpu-baseline --no-sizes --no-motion > /dev/null 2> $R/gpurun_out/sv.err
  echo "== $c ($V)"; tail -1 $R/gpurun_out/sv.err
  python3 $R/profiles/$S $R/gpurun_out/sv.bin
  rm -f $R/gpurun_out/sv.bin $R/gpurun_out/sv.bin.ext
done
