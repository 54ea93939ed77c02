#!/bin/bash
# round 5, call ah: soaks of the non-default placements of the fix-up / pack kernels (LPX_FIXUP_SIDE = 0, 1, 3) and of the
# unmasked stream pair (LPX_OVERLAP_MASK = 0) on the final tree: random LPs in random budget pieces, bit for bit
mkdir -p gpurun_out
O=gpurun_out/r05_soak_final_placements.txt
: > $O
for side in 0 1 3; do
  echo "## LPX_FIXUP_SIDE=$side, default shapes" >> $O
  LPX_FIXUP_SIDE=$side timeout -k 10 200 python scripts/soak_chain.py 50 2>&1 | tail -1 >> $O
  echo "## LPX_FIXUP_SIDE=$side, fused mid shapes" >> $O
  LPX_FIXUP_SIDE=$side timeout -k 10 200 python scripts/soak_chain.py 50 2048x4096,1024x8192 800 fused 2>&1 | tail -1 >> $O
done
echo "## LPX_OVERLAP_MASK=0 (plain streams), default shapes" >> $O
LPX_OVERLAP_MASK=0 timeout -k 10 200 python scripts/soak_chain.py 50 2>&1 | tail -1 >> $O
echo "## LPX_OVERLAP_MASK=0, fused mid shapes" >> $O
LPX_OVERLAP_MASK=0 timeout -k 10 200 python scripts/soak_chain.py 50 2048x4096,1024x8192 800 fused 2>&1 | tail -1 >> $O
cat $O
