#!/bin/bash
# round 3: is the sweep power-limited?  Board power and shader clock sampled while the micro loops the kernels.
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30
echo "--- while sweeping (np 32, real data) ---"
(timeout -k 5 60 scripts/micro/sweep_dma 32768 16384 1500 0 1 > /tmp/sw.log 2>&1 &)
sleep 6
for k in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ' '; echo; sleep 1.5; done
wait
sleep 12
cat /tmp/sw.log | grep "np 32"
