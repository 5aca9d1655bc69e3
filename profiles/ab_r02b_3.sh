cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
python -m pytest tests/test_gpu_parity.py -x -q -k "cull or hierarchy or config or synthetic or fuzz or random" > gpurun_out/gt3.log 2>&1; tail -3 gpurun_out/gt3.log
echo "== C5"; printf "RM_X=0\nRM_X=0\n" | bash profiles/ab_env.sh --config C5 --steps 20
echo "== C3"; printf "RM_X=0\n" | bash profiles/ab_env.sh --config C3
RM_LIB_PATH=$V/stamps/librusty_marcher_amd.so RM_DEBUG_STAMPS=$GRAFT_REPO_ROOT/gpurun_out/st_c5.bin python bench.py --config C5 --steps 1 --warmup 1 --no-cpu-baseline --no-sizes > /dev/null 2>gpurun_out/st_c5.err
python profiles/analyze_stamps.py gpurun_out/st_c5.bin > gpurun_out/st_c5.txt; rm -f gpurun_out/st_c5.bin; cat gpurun_out/st_c5.txt
