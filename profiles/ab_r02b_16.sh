cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
S="RM_LIB_PATH=$V/s0/librusty_marcher_amd.so"
python -m pytest tests -m gpu -x -q > gpurun_out/gt16.log 2>&1; tail -3 gpurun_out/gt16.log
echo "== C2"; printf "RM_X=0\n$S\nRM_X=0\n$S\n" | bash profiles/ab_env.sh --config C2 --steps 200
echo "== C4"; printf "RM_X=0\n$S\n" | bash profiles/ab_env.sh --config C4 --steps 20
echo "== C3"; printf "RM_X=0\n$S\n" | bash profiles/ab_env.sh --config C3
echo "== C5"; printf "RM_X=0\n" | bash profiles/ab_env.sh --config C5 --steps 20
