cd $GRAFT_REPO_ROOT
echo "== C5 cull_cos sweep"; printf "RM_X=0\nRM_CULL_COS=0.94\nRM_CULL_COS=0.96\nRM_CULL_COS=0.985\nRM_CULL_COS=0.993\nRM_X=0\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C3 cull_cos sweep"; printf "RM_X=0\nRM_CULL_COS=0.94\nRM_CULL_COS=0.99\n" | bash profiles/ab_env.sh --config C3
