cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
P="RM_LIB_PATH=$V/prev/librusty_marcher_amd.so"
echo "== C2"; printf "RM_X=0\n$P\nRM_X=0\n$P\nRM_X=0\n$P\n" | bash profiles/ab_env.sh --config C2 --steps 200
echo "== C3"; printf "RM_X=0\n$P\nRM_X=0\n$P\n" | bash profiles/ab_env.sh --config C3
echo "== C2_4K"; printf "RM_X=0\n$P\nRM_X=0\n$P\n" | bash profiles/ab_env.sh --config C2_4K --steps 40
