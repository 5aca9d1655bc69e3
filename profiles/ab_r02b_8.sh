cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
echo "== C5"; printf "RM_X=0\nRM_LIB_PATH=$V/noguard/librusty_marcher_amd.so\nRM_LIB_PATH=$V/t0/librusty_marcher_amd.so\nRM_FEEDBACK_TARGET=0\nRM_FEEDBACK_TARGET=0 RM_LIB_PATH=$V/noguard/librusty_marcher_amd.so\nRM_FEEDBACK_TARGET=0 RM_LIB_PATH=$V/t0/librusty_marcher_amd.so\nRM_FEEDBACK_TARGET=0 RM_FEEDBACK_US=35 RM_LIB_PATH=$V/noguard/librusty_marcher_amd.so\nRM_FEEDBACK_TARGET=0 RM_FEEDBACK_US=35 RM_LIB_PATH=$V/t0/librusty_marcher_amd.so\n" | bash profiles/ab_env.sh --config C5 --steps 20
