#!/bin/bash
# round 5, call u: SQ counters of decision-only launches (overlap = 0, cfg3: 32 decisions per launch, 64 workgroups = 256 waves):
# instructions per wave and decision by class, wait share
mkdir -p gpurun_out
R=$PWD
OUT=$R/gpurun_out/r05_u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_a -- python3 $R/scripts/arith_grid.py cfg3 "overlap=0" 256 64 > $OUT/pmc_a.log 2>&1; echo "pmc a rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -- python3 $R/scripts/arith_grid.py cfg3 "overlap=0" 256 64 > $OUT/pmc_b.log 2>&1; echo "pmc b rc=$?"
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python scripts/pmc_filter.py k_block_chain2_t $(f pmc_a) $OUT/decision_pmc_a.csv
python scripts/pmc_filter.py k_block_chain2_t $(f pmc_b) $OUT/decision_pmc_b.csv
python scripts/pmc_summary.py k_block_chain2_t $OUT/decision_pmc_a.csv $OUT/decision_pmc_b.csv | tee $OUT/decision_pmc_summary.txt | head -60
rm -rf $OUT/pmc_a $OUT/pmc_b
