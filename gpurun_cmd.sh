cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline > $O/write.log 2>&1
ls -la $O | head; tail -1 $O/fetch.log | cut -c1-200
