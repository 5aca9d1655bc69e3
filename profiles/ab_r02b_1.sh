cd $GRAFT_REPO_ROOT
echo "== C5"; printf 'RM_WG_WAVES=4\nRM_WG_WAVES=1\nRM_WG_WAVES=4 RM_FORCE_STACK=32\nRM_WG_WAVES=1 RM_FORCE_STACK=32\nRM_WG_WAVES=4\nRM_WG_WAVES=1\n' | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C3"; printf 'RM_WG_WAVES=4\nRM_WG_WAVES=1\nRM_WG_WAVES=4\nRM_WG_WAVES=1\nRM_WG_WAVES=1 RM_FORCE_STACK=32\n' | bash profiles/ab_env.sh --config C3
echo "== C2"; printf 'RM_X=0\nRM_FORCE_STACK=16\nRM_FORCE_STACK=32\nRM_X=0\n' | bash profiles/ab_env.sh --config C2
