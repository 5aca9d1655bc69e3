cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
L="\nRM_LIB_PATH=$V/lazy/librusty_marcher_amd.so\nRM_LIB_PATH=$V/s0/librusty_marcher_amd.so"
echo "== C2"; printf "RM_X=0$L\nRM_X=0$L\n" | bash profiles/ab_env.sh --config C2 --steps 200
echo "== C4"; printf "RM_X=0$L\n" | bash profiles/ab_env.sh --config C4 --steps 20
echo "== C2_4K"; printf "RM_X=0$L\n" | bash profiles/ab_env.sh --config C2_4K --steps 40
