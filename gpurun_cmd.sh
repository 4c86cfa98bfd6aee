cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O -o m --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 32 --no-cpu-baseline > $O/m.log 2>&1
ls $O
