cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/rusty-marcher_amd/lib/variants
echo "== C2"; printf "RM_X=0\nRM_X=0\nRM_X=0\n" | bash profiles/ab_env.sh --config C2 --steps 200
echo "== C3"; printf "RM_X=0\nRM_X=0\n" | bash profiles/ab_env.sh --config C3
echo "== C4"; printf "RM_X=0\n" | bash profiles/ab_env.sh --config C4 --steps 20
echo "== C5"; printf "RM_X=0\nRM_X=0\n" | bash profiles/ab_env.sh --config C5 --steps 20
for c in C2 C5; do
RM_LIB_PATH=$V/stamps/librusty_marcher_amd.so RM_DEBUG_STAMPS=$GRAFT_REPO_ROOT/gpurun_out/st_$c.bin python bench.py --config $c --steps 3 --warmup 3 --no-cpu-baseline --no-sizes > /dev/null 2>gpurun_out/st_$c.err
python profiles/analyze_stamps.py gpurun_out/st_$c.bin > gpurun_out/st_$c.txt; rm -f gpurun_out/st_$c.bin; cat gpurun_out/st_$c.txt
done
python -m pytest tests -m gpu -x -q > gpurun_out/gt10.log 2>&1; tail -3 gpurun_out/gt10.log
