#!/bin/bash
# round 3 evidence: the driver's command and the default command (full lines), kernel-trace stats of the driver's
# command, HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ / TCC counters of the shipped sweep kernel
# (k_sweep32_pull beside the decision kernel, the default loop).  rocprofv3 wants the program itself after "--".
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/r03_f
mkdir -p $OUT
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/drv.err | tail -1 > $OUT/bench_driver_command.json || { tail -20 $OUT/drv.err; exit 1; }
python scripts/bench_line.py drv < $OUT/bench_driver_command.json
timeout -k 10 400 python bench.py 2>$OUT/def.err | tail -1 > $OUT/bench_default_cfg4.json || { tail -20 $OUT/def.err; exit 1; }
python scripts/bench_line.py default < $OUT/bench_default_cfg4.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_drv.log 2>&1; echo "stats drv rc=$?"
ARGS="$R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --no-steady --steps 256 --warmup 64"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_TAG_STALL TCC_REQ --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"
cd $R
python scripts/pmc_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") k_sweep32_pull 32768 16384 cfg4 32 256 > $OUT/traffic_cfg4_n1.json; cat $OUT/traffic_cfg4_n1.json | tail -6
python scripts/pmc_summary.py k_sweep32_pull $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv" | sort) > $OUT/pmc_summary_k_sweep32_pull.txt; cat $OUT/pmc_summary_k_sweep32_pull.txt
find $OUT/stats_drv -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_driver_kernel_stats.csv; head -12 $OUT/bench_driver_kernel_stats.csv
