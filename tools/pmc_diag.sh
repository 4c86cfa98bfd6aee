#!/bin/bash
# Diagnostic counter passes over single conv launches (tools/conv_bench.py), one rocprofv3 --pmc pass per counter group:
#   bash tools/pmc_diag.sh <tag> <conv_bench args...>      e.g.  bash tools/pmc_diag.sh wg4 --wg4 --batch 32 res2_d27 res3_d9
# -> gpurun_out/diag_<tag>/summary.txt (per kernel: every counter, and per GRBM_GUI_ACTIVE-derived SIMD cycle)
TAG=$1; shift
OUT=$PWD/gpurun_out/diag_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/conv_bench.py --iters 3 $*"
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
G2="SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
G3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES"
G4="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"
G5="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
cd /tmp
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1))
  timeout 240 rocprofv3 --output-format csv --pmc $G --kernel-trace -d $OUT/p$i -o g -- $CMD > $OUT/p$i.log 2>&1
done
cd - > /dev/null
python tools/pmc_diag.py $OUT > $OUT/summary.txt 2>&1
tar -czf $OUT/raw.tgz -C $OUT p1 p2 p3 p4 p5 2>/dev/null; rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
cat $OUT/summary.txt
