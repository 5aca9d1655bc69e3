cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
L=""
for v in w3; do L="$L\nRM_LIB_PATH=$V/$v/librusty_marcher_amd.so"; done
echo "== C3"; printf "RM_X=0$L\nRM_X=0$L\n" | bash profiles/ab_env.sh --config C3
echo "== C5"; printf "RM_X=0$L\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C2"; printf "RM_X=0$L\n" | bash profiles/ab_env.sh --config C2 --steps 200
