#!/bin/bash
# round 4 evidence, second half (call ac): kernel stats per workload and the counter passes of the fused sweeps on the final
# kernels (k_sweep64_mfma2 second version).  rocprofv3 wants the program itself after "--".
R=$PWD
OUT=$R/gpurun_out/r04_p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel stats per workload: the driver's command on cfg4 alone and on cfg3 alone (no baselines, no extra legs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cfg3 --no-steady --no-fused --no-onepass --no-parity > $OUT/stats_cfg4.log 2>&1; echo "stats cfg4 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg3 --no-cpu-baseline --no-onepass --no-parity > $OUT/stats_cfg3.log 2>&1; echo "stats cfg3 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fused -- python3 $R/bench.py --gpus 1 --steps 512 --warmup 64 --no-cpu-baseline --no-cfg3 --no-steady --no-onepass --no-parity --option fused=1 > $OUT/stats_fused.log 2>&1; echo "stats fused rc=$?"
F32="$R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --no-steady --no-onepass --steps 256 --warmup 64 --option fused=1 --option block=32"
F64="$R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --no-steady --no-onepass --steps 256 --warmup 64 --option fused=1"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc1_f32 -- python3 $F32 > $OUT/pmc1_f32.log 2>&1; echo "pmc1 f32 rc=$?"
rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc1_f64 -- python3 $F64 > $OUT/pmc1_f64.log 2>&1; echo "pmc1 f64 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_f32 -- python3 $F32 > $OUT/fetch_f32.log 2>&1; echo "fetch f32 rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_f32 -- python3 $F32 > $OUT/write_f32.log 2>&1; echo "write f32 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_f64 -- python3 $F64 > $OUT/fetch_f64.log 2>&1; echo "fetch f64 rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_f64 -- python3 $F64 > $OUT/write_f64.log 2>&1; echo "write f64 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2_f64 -- python3 $F64 > $OUT/pmc2_f64.log 2>&1; echo "pmc2 f64 (MFMA counters; may not exist) rc=$?"
cd $R
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python scripts/pmc_summary.py k_sweep32_pull $(f pmc1_f32) > $OUT/pmc_summary_fused_k_sweep32_pull.txt; cat $OUT/pmc_summary_fused_k_sweep32_pull.txt | head -30
python scripts/pmc_summary.py k_sweep64_mfma2 $(f pmc1_f64) $(f pmc2_f64) > $OUT/pmc_summary_fused_k_sweep64_mfma2.txt; cat $OUT/pmc_summary_fused_k_sweep64_mfma2.txt | head -40
python scripts/pmc_traffic.py $(f fetch_f32) $(f write_f32) k_sweep32_pull 32768 16384 cfg4 32 256 > $OUT/traffic_cfg4_fused_block32.json; tail -5 $OUT/traffic_cfg4_fused_block32.json
python scripts/pmc_traffic.py $(f fetch_f64) $(f write_f64) k_sweep64_mfma2 32768 16384 cfg4 64 256 > $OUT/traffic_cfg4_fused_block64.json; tail -5 $OUT/traffic_cfg4_fused_block64.json
for W in cfg4 cfg3 fused; do find $OUT/stats_$W -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$W.csv; done
head -6 $OUT/kernel_stats_cfg4.csv | cut -c1-200
# keep the merge small: drop the raw traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
