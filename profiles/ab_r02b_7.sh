cd $GRAFT_REPO_ROOT
echo "== C5"; printf "RM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK_TARGET=0\nRM_X=0\nRM_FEEDBACK_TARGET=16384\nRM_FEEDBACK_TARGET=0 RM_FEEDBACK_US=70\nRM_FEEDBACK_TARGET=0 RM_FEEDBACK_US=35\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C4"; printf "RM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK=1\n" | bash profiles/ab_env.sh --config C4 --steps 20
echo "== C2"; printf "RM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK=1\n" | bash profiles/ab_env.sh --config C2
python -m pytest tests/test_gpu_parity.py -x -q -k "feedback" > gpurun_out/gt7.log 2>&1; tail -3 gpurun_out/gt7.log
