"""The pure helpers behind bench.py's JSON line: the bounded roofline of one sweep launch and which half of the blocked
loop sets the pace (no GPU, no library)."""
import math

import bench


def test_roofline_block_is_bounded_and_names_the_binding_term():
    # cfg4, one sweep of 32 pivots in 1.72 ms: the HBM term (16 m n / 8 TB/s = 1.074 ms) is the larger lower bound
    r = bench.roofline_block(32768, 16384, 32, 1.72, "k_sweep32_pull", 16)
    assert r["bound"] == "hbm" and 0.62 < r["frac"] < 0.63 and r["frac"] <= 1.0
    assert math.isclose(r["algorithmic_bytes_per_launch"], 16.0 * 32768 * 16384)
    assert math.isclose(r["pivot_equiv_frac"], 32 * r["hbm_frac"])   # SURVEY 8(d)'s per-pivot figure
    # 64 pivots per launch: 2 m n K / 39.3 T op/s = 1.749 ms > 1.074 ms, the fp64 term binds
    r64 = bench.roofline_block(32768, 16384, 64, 3.30, "k_sweep64_pull", 8)
    assert r64["bound"] == "fp64_valu" and r64["frac"] <= 1.0 and r64["unit"] == "T fp64 instr/s"
    # the same sweep of 32 at the clock an fp64-dense kernel really holds (1.56 GHz): the instruction term is the larger
    # one there (2 m n K / (256 x 4 x 16 x 1.56 GHz) = 1.345 ms > 1.074 ms) although `frac` stays quoted against the
    # data-sheet peaks; cycles per launch = time x clock
    rc = bench.roofline_block(32768, 16384, 32, 1.80, "k_sweep32_pull", 16, clock_mhz=1560)
    assert rc["bound"] == "fp64_valu" and rc["bound_at"] == "measured clock, 256 CUs" and math.isclose(rc["frac"], r["frac"] * 1.72 / 1.80)
    assert math.isclose(rc["cycles_per_launch"], 1.80e-3 * 1.56e9) and 1.34 < rc["lower_bound_ms"]["fp64_valu_at_clock"] < 1.35
    # fused arithmetic: one instruction per entry and pivot, the memory pass binds again at that clock
    rf = bench.roofline_block(32768, 16384, 32, 1.54, "k_sweep32_pull", 16, fused=True, clock_mhz=1900)
    assert rf["bound"] == "hbm" and 0.69 < rf["frac"] < 0.70 and rf["arithmetic"].startswith("fused")
    # blocks of 64 on the matrix cores beside the decisions (round 4's driver record: 1.898 ms at 2.048 GHz on the 192 CUs
    # the decisions leave): 16 x 64 cycles per 16 x 16 tile = m n 64 lane-operations on 192 x 4 SIMDs = 1.365 ms > the memory
    # pass (1.074 ms): the matrix pipe binds, although on all 256 CUs it would not (Weak 4)
    rm = bench.roofline_block(32768, 16384, 64, 1.898, "k_sweep64_mfma2", 8, fused=True, clock_mhz=2048, cus=192)
    assert rm["bound"] == "mfma" and rm["cus"] == 192 and 1.36 < rm["lower_bound_ms"]["fp64_on_its_cus_at_clock"] < 1.37
    assert bench.roofline_block(32768, 16384, 64, 1.898, "k_sweep64_mfma2", 8, fused=True, clock_mhz=2048)["bound"] == "hbm"
    # nothing sampled: no fraction is invented
    assert bench.roofline_block(8192, 16384, 32, float("nan"), "k", 0)["frac"] is None


def test_traffic_files_match_their_kernel_and_mode():
    """roofline.traffic is quoted only from a counter file of the same workload, kernel and arithmetic mode."""
    t, src = bench.load_traffic("cfg4", 1, 64, "k_sweep64_mfma2", fused=True)
    assert t and "k_sweep64_mfma2" == src["kernel"] and 1.0 < t / (16.0 * 32768 * 16384) < 1.06
    assert bench.load_traffic("cfg4", 1, 64, "k_sweep64_mfma2", fused=False) == (None, None)
    # the driver's 20-pivot command in the by-size arithmetic: the handle's block size is 64, the kernel that ran is the
    # sweep of blocks of up to 32 in the fused mode
    t32, src32 = bench.load_traffic("cfg4", 1, 64, "k_sweep32_pull", fused=True)
    assert t32 and src32["file"].endswith("fused_block32.json") and src32["kernel"] == "k_sweep32_pull"
    assert bench.load_traffic("cfg4", 1, 32, "k_sweep64_one", fused=True) == (None, None)


def test_loop_bound_names_the_slower_half():
    # cfg4 steady: a block of 32 pivots every 1.945 ms, its sweep 1.786 ms -> the sweep sets the pace
    assert bench.loop_bound(1.945 / 32, 32, 1.786, {"overlapped": 1})["bound"] == "sweep"
    # cfg3 steady: 0.665 ms per block, sweep 0.484 ms -> the decisions do
    assert bench.loop_bound(0.665 / 32, 32, 0.484, {"overlapped": 1})["bound"] == "decisions"
    # cfg4, fused mode, blocks of 64: the sweep 1.895 ms, the block 2.32 ms: the fix-up and the pack behind the sweep
    lb = bench.loop_bound(2.32 / 64, 64, 1.895, {"overlapped": 1})
    assert lb["bound"] == "sweep + fix-up" and abs(lb["other_ms"] - 0.425) < 1e-9
    # a budget of one block (the driver's 20 pivots): nothing runs side by side
    assert bench.loop_bound(1.87 / 20, 20, 1.40, {"overlapped": 0})["bound"].startswith("serial")
    # one pass per pivot has no second half
    assert bench.loop_bound(1.36, 1, 1.36, None) is None
