cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
for c in C2 C3 C5 C2_4K; do
 st=100; [ $c = C5 ] && st=20; [ $c = C2_4K ] && st=40
 echo "== $c"; printf "RM_X=0\nRM_LIB_PATH=$V/tile8/librusty_marcher_amd.so\nRM_X=0\nRM_LIB_PATH=$V/tile8/librusty_marcher_amd.so\n" | bash profiles/ab_env.sh --config $c --steps $st
done
