"""Print the phase durations of k_block_chain's last block (LPX_CHAIN_TRACE file; 100 MHz ticks -> us)."""
import sys
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
rows = [r for r in rows if r[1]]
print("s  phaseA  barrier1  phaseB  barrier2  total(us)")
prev = None
for r in rows:
    k, t0, t1, t2, t3, t4 = r
    print("%2d %7.2f %8.2f %7.2f %8.2f %8.2f" % (k, (t1 - t0) / 100, (t2 - t1) / 100, (t3 - t2) / 100, (t4 - t3) / 100, (t4 - t0) / 100))
if rows:
    print("block total %.1f us for %d decisions" % ((rows[-1][5] - rows[0][1]) / 100, len(rows)))
