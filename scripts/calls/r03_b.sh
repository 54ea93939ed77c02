#!/bin/bash
# round 3: where the LDS-DMA sweep kernel's time goes — the memory pass alone, the arithmetic alone, the streaming
# ceiling with fp64 work riding on it (stream_fp64), and SQ / TCC counters of both steady-state kernels (micro, cfg4)
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/r03_b
mkdir -p $OUT
echo "== stream_fp64"; timeout -k 10 200 scripts/micro/stream_fp64 4 10 | tee $OUT/stream_fp64.txt
echo "== diag 1: memory pass alone"; timeout -k 10 120 scripts/micro/sweep_dma_diag1 32768 16384 10 0 1 | grep "np 32" 
echo "== diag 2: arithmetic alone"; timeout -k 10 120 scripts/micro/sweep_dma_diag2 32768 16384 10 0 1 | grep "np 32"
echo "== full"; timeout -k 10 120 scripts/micro/sweep_dma 32768 16384 10 0 1 | grep "np 32"
cd /tmp && export TMPDIR=/tmp
M="$R/scripts/micro/sweep_dma 32768 16384 4 0 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- $M > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/pmc2 -- $M > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc3 -- $M > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_TAG_STALL TCC_REQ --output-format csv -d $OUT/pmc4 -- $M > $OUT/pmc4.log 2>&1; echo "pmc4 rc=$?"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc5 -- $M > $OUT/pmc5.log 2>&1; echo "pmc5 rc=$?"
cd $R
for k in k_sweep32_steady k_sweep32_dma; do
  python scripts/pmc_summary.py $k $(find $OUT -name "*counter_collection.csv" | sort) > $OUT/pmc_summary_$k.txt
  cat $OUT/pmc_summary_$k.txt
done
