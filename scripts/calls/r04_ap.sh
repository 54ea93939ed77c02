#!/bin/bash
# round 4, call ap: the fix-up's scattered 8-byte column writes as non-temporal stores (experiment): k_block_fixup's
# duration in the fused loop with blocks of 64 at cfg4 and in the default loop at cfg4 / cfg3
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in default fixnt; do
  if [ $L = fixnt ]; then export LPX_LIB_PATH=$R/gpurun_variants/liblpx_fixnt.so; fi
  for W in "cfg4 fused=1" "cfg4 fused=0" "cfg3 fused=1"; do
    set -- $W
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_ap_${L}_$1_$2 -- python3 $R/scripts/arith_grid.py $1 "$2" 320 64 > $R/gpurun_out/r04_ap.log 2>&1
    F=$(find $R/gpurun_out/r04_ap_${L}_$1_$2 -name "*kernel_stats.csv" | head -1)
    echo "== $L $1 $2: $(grep pivots/s $R/gpurun_out/r04_ap.log | cut -c1-110)"
    grep "k_block_fixup" $F | awk -F, '{print "   k_block_fixup calls " $2 " avg ns " $4 " max " $7}'
    find $R/gpurun_out/r04_ap_${L}_$1_$2 -name "*kernel_trace.csv" -delete; find $R/gpurun_out/r04_ap_${L}_$1_$2 -name "*agent_info.csv" -delete
  done
done
