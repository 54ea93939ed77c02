#!/bin/bash
# round 5, final evidence (call p): kernel stats per workload on HEAD, the counter passes of the default loop's sweep at cfg4
# (by-size arithmetic = fused: k_sweep64_mfma2), every summary made from the very CSVs that are kept (filtered to that kernel),
# and the two bench lines.  rocprofv3 wants the program itself after "--".
R=$PWD
OUT=$R/gpurun_out/r05_p
mkdir -p $OUT $OUT/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cfg3 --no-steady --no-fused --no-onepass --no-parity > $OUT/stats_cfg4.log 2>&1; echo "stats cfg4 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg3 --no-cpu-baseline --no-onepass --no-parity > $OUT/stats_cfg3.log 2>&1; echo "stats cfg3 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_steady -- python3 $R/bench.py --gpus 1 --steps 512 --warmup 64 --no-cpu-baseline --no-cfg3 --no-steady --no-onepass --no-parity > $OUT/stats_steady.log 2>&1; echo "stats steady rc=$?"
F64="$R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --no-steady --no-onepass --steps 256 --warmup 64"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc1 -- python3 $F64 > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $F64 > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $F64 > $OUT/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $F64 > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
for P in pmc1 pmc2 fetch write; do python scripts/pmc_filter.py k_sweep64_mfma2 $(f $P) $OUT/pmc/${P}_k_sweep64_mfma2.csv; done
mkdir -p profiles/r05_pmc && cp $OUT/pmc/*.csv profiles/r05_pmc/
python scripts/pmc_summary.py k_sweep64_mfma2 profiles/r05_pmc/pmc1_k_sweep64_mfma2.csv profiles/r05_pmc/pmc2_k_sweep64_mfma2.csv > $OUT/pmc_summary_k_sweep64_mfma2.txt; head -40 $OUT/pmc_summary_k_sweep64_mfma2.txt
python scripts/pmc_traffic.py profiles/r05_pmc/fetch_k_sweep64_mfma2.csv profiles/r05_pmc/write_k_sweep64_mfma2.csv k_sweep64_mfma2 32768 16384 cfg4 64 256 > $OUT/traffic_cfg4_block64.json; tail -6 $OUT/traffic_cfg4_block64.json
for W in cfg4 cfg3 steady; do find $OUT/stats_$W -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$W.csv; done
head -8 $OUT/kernel_stats_steady.csv | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/fetch $OUT/write
# the two bench lines (the driver's command, and the default command: steady-state leg, the other arithmetic mode, cfg3, baselines)
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "driver bench rc=$?"
python scripts/bench_line.py < $OUT/bench_driver_command.json
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
python scripts/bench_line.py < $OUT/bench_default.json
