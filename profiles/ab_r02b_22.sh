cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
python -m pytest tests -m gpu -x -q > gpurun_out/gt22.log 2>&1; tail -3 gpurun_out/gt22.log
L="\nRM_LIB_PATH=$V/hand/librusty_marcher_amd.so"
echo "== C5 (second line of a pair: -DRM_HANDOVER=1)"; printf "RM_X=0$L\nRM_X=0$L\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C3"; printf "RM_X=0$L\nRM_X=0$L\n" | bash profiles/ab_env.sh --config C3
echo "== C2"; printf "RM_X=0$L\n" | bash profiles/ab_env.sh --config C2 --steps 200
