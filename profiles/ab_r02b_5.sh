cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "cull or hierarchy or config or synthetic or fuzz or random or sweep" > gpurun_out/gt5.log 2>&1; tail -3 gpurun_out/gt5.log
echo "== C5"; printf "RM_WALK_COS=-2\nRM_X=0\nRM_WALK_COS=-2\nRM_X=0\nRM_WALK_COS=0.9\nRM_WALK_COS=0.99\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C3"; printf "RM_X=0\n" | bash profiles/ab_env.sh --config C3
