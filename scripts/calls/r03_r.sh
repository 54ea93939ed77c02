#!/bin/bash
# round 3: shader CYCLES per decision (s_memtime stamps, diagnostic build) alone and beside the sweep, next to the
# microseconds of the normal build: is a decision beside the sweep slower in cycles, or only in time (clock)?
echo "== microseconds (100 MHz stamps): alone / beside the sweep"
timeout -k 10 100 python scripts/chain_trace.py cfg3 256 overlap=0 2>&1 | tail -1
timeout -k 10 100 python scripts/chain_trace.py cfg3 256 2>&1 | tail -1
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_clk.so
echo "== shader cycles / 100 (s_memtime stamps): alone / beside the sweep"
timeout -k 10 100 python scripts/chain_trace.py cfg3 256 overlap=0 2>&1 | tail -1
timeout -k 10 100 python scripts/chain_trace.py cfg3 256 2>&1 | tail -1
