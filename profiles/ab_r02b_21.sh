cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
RM_LIB_PATH=$V/split/librusty_marcher_amd.so python -m pytest tests/test_gpu_parity.py -x -q -k "cull or hierarchy or config or synthetic or fuzz or random or sweep or feedback" > gpurun_out/gt21.log 2>&1; tail -3 gpurun_out/gt21.log
L="\nRM_LIB_PATH=$V/split0/librusty_marcher_amd.so\nRM_LIB_PATH=$V/split/librusty_marcher_amd.so"
echo "== C5"; printf "RM_X=0$L\nRM_X=0$L\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C3"; printf "RM_X=0$L\n" | bash profiles/ab_env.sh --config C3
