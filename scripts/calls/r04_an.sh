#!/bin/bash
# round 4, final evidence (call an): decision traces of the release library (8 stamps, chain_trace_fine.py reads them too),
# kernel stats per workload, counter passes of the fused sweeps.  rocprofv3 wants the program itself after "--".
R=$PWD
mkdir -p gpurun_out
O=gpurun_out/r04_an_traces.txt
: > $O
for W in cfg3 cfg4; do
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 2>&1 | tail -3 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 2>&1 | tail -3 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 fused=1 2>&1 | tail -3 >> $O
done
cat $O
bash scripts/calls/r04_ac.sh
