cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "feedback" > gpurun_out/gt6.log 2>&1; tail -5 gpurun_out/gt6.log
echo "== C5"; printf "RM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK_TARGET=4096\nRM_FEEDBACK_TARGET=16384\nRM_FEEDBACK_TARGET=2048\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C4"; printf "RM_FEEDBACK=0\nRM_X=0\nRM_FEEDBACK=0\nRM_X=0\n" | bash profiles/ab_env.sh --config C4 --steps 20
echo "== C2"; printf "RM_FEEDBACK=0\nRM_FEEDBACK=1\nRM_FEEDBACK=0\nRM_FEEDBACK=1\n" | bash profiles/ab_env.sh --config C2
echo "== C2_4K"; printf "RM_FEEDBACK=0\nRM_FEEDBACK=1\nRM_FEEDBACK=0\nRM_FEEDBACK=1\n" | bash profiles/ab_env.sh --config C2_4K --steps 40
echo "== C3"; printf "RM_FEEDBACK=0\nRM_FEEDBACK=1\n" | bash profiles/ab_env.sh --config C3
RM_FEEDBACK=1 python -m pytest tests -m gpu -x -q > gpurun_out/gt6_all.log 2>&1; tail -3 gpurun_out/gt6_all.log
