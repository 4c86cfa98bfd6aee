set -x
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1/bench.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof1/bench.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head -20
