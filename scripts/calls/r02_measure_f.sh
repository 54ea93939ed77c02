#!/bin/bash
# PMC passes on the blocked sweep (cfg4, K = 32, alone and in place) + kernel-trace stats of the default bench and of
# the one-pass cfg3 loop.  rocprofv3 wants the program itself after "--".
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/prof_f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 128 --warmup 32 --option overlap=0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM TCC_EA0_WRREQ TCC_TAG_STALL --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1; echo "pmc4 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $R/bench.py --no-cpu-baseline --no-cfg3 --no-parity > $OUT/stats_cfg4.log 2>&1; echo "stats cfg4 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3_onepass -- python3 $R/bench.py --workload cfg3 --no-cpu-baseline --no-parity --option block=1 --steps 200 --warmup 10 > $OUT/stats_cfg3_onepass.log 2>&1; echo "stats cfg3 rc=$?"
find $OUT -name "*.csv" | head -30
