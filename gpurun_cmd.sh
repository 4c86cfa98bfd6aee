cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final_b32
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 32 > $O/bench.log 2>&1
O1=$GRAFT_REPO_ROOT/gpurun_out/final_b1
mkdir -p $O1
rocprofv3 --kernel-trace --stats -d $O1 -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 1 --no-cpu-baseline > $O1/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.json | cut -c1-300
