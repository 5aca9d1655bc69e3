cd $GRAFT_REPO_ROOT
echo "== C5"; printf "RM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK_TARGET=8192\nRM_FEEDBACK_TARGET=12288\nRM_FEEDBACK_TARGET=16384\nRM_FEEDBACK_TARGET=0\nRM_X=0\n" | bash profiles/ab_env.sh --config C5 --steps 20
python -m pytest tests/test_gpu_parity.py -x -q -k "feedback" > gpurun_out/gt9.log 2>&1; tail -3 gpurun_out/gt9.log
