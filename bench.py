#!/usr/bin/env python3
"""bench.py — simplex pivots/s of the MI355X-native pivot loop, with its roofline, an oracle replay and CPU baselines.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one simplex pivot (entering scan + ratio test + tableau update + index swap, reference
LPState.java:274-320, :133-181) of a dense random LP:  A ~ U(0,1), b = (n/4) U(1,2), c ~ U(0,1), maximise
(SURVEY §8d).  The default workload is BASELINE cfg4 (m=32768, n=16384, 4 GiB fp64 tableau): it is the
configuration the metric's 1/2/4/8-GPU scaling is quoted on and it fits one GPU, so every N runs the SAME
job ("scaling": "strong").  At N>1 the tableau is cut into row blocks behind ONE lpx_multi handle (C ABI): rank 0
drives all N GPUs, their persistent decision kernels exchange candidates and pivot rows by direct xGMI stores, the
other ranks only join the barriers (`--multi-backend rccl`: round 1's one process per GPU + RCCL all_gather).

The timed region starts with the tableau resident in HBM (upload excluded).  Rank 0 prints ONE JSON line.
`roofline`: the sweep / row-update launch is bounded below by one 16*m_local*n-byte pass at 8 TB/s and by its
2*m_local*n*K unfusable fp64 operations at 39.3 T op/s; frac = the larger bound / the mean launch time measured with
HIP events on the launch stream inside the timed region (<= 1), `bound` names the larger term (roofline_block).
`parity_after_timed_region`: warm-up + steps pivots replayed on the fp64 oracle and compared bit for bit with what the
timed handle holds.  `cpu_baseline` = the decimal-15 oracle (the reference's BigDecimal arithmetic, 4 threads as in
pivotConcurrently) and `cpu_baseline_fp64` = the fp64 oracle on all host cores, both timed here on a bounded
row-sample of the same tableau.  At N=1 the line also carries a `cfg3` object (BASELINE's single-GPU roofline config).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfg2": (1024, 2048),     # cache-resident; parity config, not a roofline config
    "cfg3": (8192, 16384),    # 1 GiB tableau, 2.147 GB algorithmic bytes per pivot
    "cfg4": (32768, 16384),   # 4 GiB tableau, 8.59 GB per pivot; the 1/2/4/8-GPU scaling config
    "cfg4_shard8": (4096, 16384),   # what ONE rank streams per pivot when cfg4 is sharded over 8 GPUs
    "cfg4_shard2": (16384, 16384),
}
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290
# fp64 vector rate for UNFUSED operations: 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2.4 GHz = 39.3 T operations/s.
# (The 78.6 TFLOP/s vector peak of the data sheet counts an FMA as two; this path must round the product and the
# difference separately — reference LPState.java:162 — so a multiply and a subtract are one operation each.)
FP64_VALU_PEAK_TFLOPS = 39.3
CHUNK = 1024


MFMA_KERNELS = ("k_sweep64_mfma", "k_sweep64_mfma2")


def roofline_block(m_local, n, pivots_per_launch, avg_ms, kernel, launches, traffic=None, traffic_source=None,
                   fused=False, clock_mhz=0, cus=0):
    """The bounded roofline figure of one row-update / sweep launch.
    One launch reads and writes every fp64 entry once (16*m*n bytes — counter-verified, profiles/) and issues, per entry
    and pivot it applies, TWO fp64 instructions (v_mul_f64 + v_add_f64: the default arithmetic rounds the product and the
    difference separately, LPState.java:162) or ONE (v_fma_f64, the opt-in fused mode).  Its time is bounded below by
        t_hbm = 16*m*n / 8 TB/s     and     t_valu = instr*m*n*pivots_per_launch / (256 CUs x 4 SIMDs x 16 lanes x clock)
    `frac` = max(t_hbm, t_valu at the NOMINAL 2.4 GHz = 39.3 T instr/s) / measured mean launch time: against the data-sheet
    peaks, <= 1 by construction.  `bound` names the larger of t_hbm and t_valu AT THE CLOCK THE CHIP HELD during the sweep
    (`clock_ghz`: in-kernel s_memtime against the 100 MHz counter, lpx_state_info.sweep_clock_mhz; nominal when it was not
    measured): under the 1400 W package cap an fp64-dense sweep runs at 1.5-2.0 GHz and the instruction term is then the
    larger one long before it is at 2.4 GHz.  `cycles_per_launch` = launch time x that clock (a kernel at its
    instruction-issue floor costs the same cycles whatever the clock).  The instruction term of `bound` is taken on the CUs
    the sweep's stream really has (`cus`: lpx_state_info.sweep_cus — in the overlapped loop the decisions keep 4 or 8 CUs
    per XCD to themselves); on the matrix cores (k_sweep64_mfma2: fp64 MFMA rate = fp64 vector rate on MI355X) the name
    is "mfma".  `pivot_equiv_frac` keeps SURVEY 8(d)'s per-PIVOT figure (16*m*n bytes per pivot / 8 TB/s), which
    exceeds 1 when one sweep applies several pivots."""
    if not launches or not (avg_ms > 0):
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": traffic,
                "traffic_source": traffic_source, "kernel": kernel, "launches_sampled": launches}
    t = avg_ms * 1e-3
    bytes_moved = 16.0 * m_local * n
    ipe = 1.0 if fused else 2.0                       # fp64 VALU instructions per entry and pivot
    flops = ipe * m_local * n * pivots_per_launch     # lane-instructions
    t_hbm = bytes_moved / (HBM_PEAK_GBS * 1e9)
    t_valu = flops / (FP64_VALU_PEAK_TFLOPS * 1e12)
    clock_ghz = clock_mhz / 1e3 if clock_mhz and clock_mhz > 0 else None
    ncu = int(cus) if cus and cus > 0 else 256
    t_valu_clk = flops / (ncu * 4 * 16 * (clock_ghz or 2.4) * 1e9)   # on the sweep's CUs, at the clock held
    hbm_peak_bound = t_hbm >= t_valu                  # which data-sheet peak `frac` / `achieved` are quoted against
    fp64_name = "mfma" if fused and kernel in MFMA_KERNELS else "fp64_valu"   # ("mfma": the fp64 matrix pipe, v_mfma_f64_16x16x4)
    out = {"bound": "hbm" if t_hbm >= t_valu_clk else fp64_name,
           "bound_at": ("measured clock" if clock_ghz else "nominal clock (not measured)") + ", %d CUs" % ncu,
           "cus": ncu,
           "achieved": bytes_moved / t / 1e9 if hbm_peak_bound else flops / t / 1e12,
           "peak": HBM_PEAK_GBS if hbm_peak_bound else FP64_VALU_PEAK_TFLOPS,
           "unit": "GB/s" if hbm_peak_bound else "T fp64 instr/s",
           "frac": max(t_hbm, t_valu) / t,
           "traffic": traffic, "traffic_source": traffic_source,
           "kernel": kernel, "avg_kernel_ms": avg_ms, "launches_sampled": launches,
           "pivots_per_launch": pivots_per_launch,
           "arithmetic": ("fused (four pivots per v_mfma_f64_16x16x4 = a chain of four fused multiply-adds per entry)"
                          if fused and kernel in ("k_sweep64_mfma", "k_sweep64_mfma2") else "fused (one v_fma_f64 per entry and pivot)" if fused
                          else "two roundings (v_mul_f64 + v_add_f64 per entry and pivot)"),
           "clock_ghz": clock_ghz,
           "cycles_per_launch": t * clock_ghz * 1e9 if clock_ghz else None,
           "lower_bound_ms": {"hbm": 1e3 * t_hbm, "fp64_valu": 1e3 * t_valu, "fp64_valu_at_clock": 1e3 * t_valu_clk,
                              "fp64_on_its_cus_at_clock": 1e3 * t_valu_clk},
           "hbm_GBps": bytes_moved / t / 1e9, "hbm_frac": t_hbm / t,
           "fp64_valu_Tinstr": flops / t / 1e12, "fp64_valu_frac": t_valu / t,
           "fp64_valu_frac_at_clock": t_valu_clk / t,
           "algorithmic_bytes_per_launch": bytes_moved,
           "pivot_equiv_GBps": bytes_moved * pivots_per_launch / t / 1e9,
           "pivot_equiv_frac": bytes_moved * pivots_per_launch / t / 1e9 / HBM_PEAK_GBS}
    if fp64_name == "mfma":
        # the same launch against the dense fp64 MFMA peak (a multiply-add = 2 flop; v_mfma_f64_16x16x4 issues in 64 cycles
        # per SIMD: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz = 78.6 TFLOP/s, the vector peak), and against what its own CUs
        # give at the clock the chip held — the figure that says how far the kernel is from its binding pipe
        tf = 2.0 * flops / t / 1e12
        peak_here = ncu * 4 * 32 * (clock_ghz or 2.4) * 1e9 / 1e12
        out["mfma"] = {"achieved": tf, "peak": 2.0 * FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / (2.0 * FP64_VALU_PEAK_TFLOPS),
                       "peak_on_its_cus_at_clock": peak_here, "frac_on_its_cus_at_clock": tf / peak_here}
    return out


def loop_bound(ms_per_step, pivots_per_launch, avg_kernel_ms, info=None):
    """Which of the two concurrent halves of the blocked loop sets the pace: one sweep launch applies
    `pivots_per_launch` pivots while the decisions of the next block are taken beside it, so a block costs
    max(decisions, sweep).  "decisions" when the block takes 10 % longer than its sweep (the sweep kernel then idles
    part of the time and its roofline fraction says nothing about the loop), else "sweep".  Between the two, a block up to
    25 % longer than its sweep is what the sweep's own stream adds behind it — the fix-up of the block's rows and columns
    (0.2 ms for a block of 64 at cfg4), the multiplier pack, launch gaps — and the first block's decisions, which nothing
    hides, spread over the call: "sweep + fix-up".  `other_ms` = block - sweep."""
    try:
        if not (avg_kernel_ms > 0) or not (pivots_per_launch >= 2):
            return None
        block_ms = ms_per_step * pivots_per_launch
        if info is not None and not info.get("overlapped", 1):   # a budget of one block: nothing runs side by side
            return {"block_ms": block_ms, "sweep_ms": avg_kernel_ms, "bound": "serial: the decisions, then one sweep"}
        ratio = block_ms / avg_kernel_ms
        return {"block_ms": block_ms, "sweep_ms": avg_kernel_ms, "other_ms": block_ms - avg_kernel_ms,
                "bound": "sweep" if ratio <= 1.1 else ("sweep + fix-up" if ratio <= 1.25 else "decisions")}
    except Exception:
        return None


def load_traffic(workload, world, block, kernel, fused=False):
    """PMC-measured HBM bytes per launch (profiles/*traffic_*.json, scripts/pmc_traffic.py) and where the figure comes
    from — only when a file was measured for this workload, GPU count, kernel AND arithmetic mode (a sweep launch moves the
    tableau once whatever the number of pivots it applies, so the kernel decides, not the block size set on the handle);
    otherwise (None, None): the number is a property of another run (TCC counters cannot be read from inside bench.py)."""
    names = []
    if world == 1 and fused:
        names += ["r05_traffic_%s_block64.json" % workload, "r04_traffic_%s_fused_block%d.json" % (workload, int(block)),
                  "r04_traffic_%s_fused_block32.json" % workload, "r04_traffic_%s_fused_block64.json" % workload]
    if not fused:
        names += ["traffic_%s_n%d.json" % (workload, world)]
    for name in names:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            if t.get("kernel") == kernel:
                return t.get("hbm_bytes_per_launch"), {"file": "profiles/" + name, "kernel": t.get("kernel"),
                                                       "steps": t.get("steps"), "date": t.get("date"),
                                                       "pivots_per_sweep": t.get("pivots_per_sweep"),
                                                       "note": "rocprofv3 --pmc passes of this command on another box"}
        except Exception:
            pass
    return None, None


def kernel_label(block, info, ld_is_whole_strips=True):
    """The kernel the event-bracketed interval covers, as rocprofv3 names it (lpx_state_get_info)."""
    if block == 1:
        return "k_update"
    name = (info or {}).get("sweep_kernel_name") or "k_update_multi"
    return name


def parity_after(st, A, b, c, pivots, m, n, threads, max_pivots, fused=False):
    """Replays `pivots` pivots of the same LP on the fp64 oracle (the checker, not the thing measured; its fused
    instantiation for a handle in the fused-arithmetic mode) and compares what the timed handle holds now: v, perm, b, c
    bit for bit and the position-keyed checksum of the tableau."""
    if pivots > max_pivots:
        return {"checked": False, "reason": "%d pivots to replay > --parity-max-pivots %d" % (pivots, max_pivots)}
    from linear_programming_solver_amd.lp_state import checksum_host
    from oracle import pyoracle as orc
    orc.build()
    t0 = time.perf_counter()
    ref = orc.State(A, b, c, kind=orc.FP64_FUSED if fused else orc.FP64, with_perm=True)
    r = ref.simplex_loop(max_pivots=pivots, threads=threads)
    wA, wb, wc, wv, wperm = ref.read()
    ref.close()
    _, gb, gc, gv, gperm = st.read(want_A=False)
    mask = (1 << 64) - 1
    sa = 0
    for r0 in range(0, m, 2048):
        sa = (sa + checksum_host(wA[r0:r0 + 2048], wb[:0], wc[:0], row0=r0)[0]) & mask
    del wA
    same = {"v": bool(np.float64(gv).view(np.uint64) == np.float64(wv).view(np.uint64)),
            "perm": bool(np.array_equal(gperm, wperm)),
            "b": bool(np.array_equal(gb.view(np.uint64), wb.view(np.uint64))),
            "c": bool(np.array_equal(gc.view(np.uint64), wc.view(np.uint64))),
            "A_checksum": bool(st.checksum()[0] == sa)}
    return {"checked": True, "ok": all(same.values()) and r["pivots"] == pivots, "pivots_replayed": int(r["pivots"]),
            "bit_identical": same,
            "oracle": "%s restatement, %d threads" % ("fp64 fused-multiply-add" if fused else "fp64", threads),
            "seconds": time.perf_counter() - t0}


def gen_rows(m, n, seed, r0, r1):
    """Rows [r0, r1) of the global synthetic LP, identical for every sharding (per-1024-row-chunk streams)."""
    A = np.empty((r1 - r0, n), dtype=np.float64)
    b = np.empty(r1 - r0, dtype=np.float64)
    first = r0 // CHUNK
    last = (r1 - 1) // CHUNK if r1 > r0 else first - 1
    for ch in range(first, last + 1):
        rng = np.random.default_rng([seed, 1, ch])
        rows = min(CHUNK, m - ch * CHUNK)
        blk = rng.random((rows, n))
        bb = (n / 4.0) * (1.0 + rng.random(rows))
        lo, hi = max(r0, ch * CHUNK), min(r1, ch * CHUNK + rows)
        A[lo - r0:hi - r0] = blk[lo - ch * CHUNK:hi - ch * CHUNK]
        b[lo - r0:hi - r0] = bb[lo - ch * CHUNK:hi - ch * CHUNK]
    c = np.random.default_rng([seed, 2]).random(n)
    return A, b, c


def host_cores():
    """CPU threads this process may really use: affinity mask and cgroup quota, not the host's core count
    (a GPU box gives one job a 16-CPU share of a much larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("LPX_BENCH_CPU_THREADS", "16"))))


def java_baseline(m_full, n, budget_s):
    """if a JDK exists on this box, time bench_java/PivotBench.java (BigDecimal, 4 threads,
    the reference's pivotConcurrently structure) at cfg2 size and scale by element count.  The build image has
    no JDK, so this normally reports unavailable."""
    import shutil
    import subprocess
    import tempfile
    javac, java = shutil.which("javac"), shutil.which("java")
    home = os.environ.get("JAVA_HOME")
    if (not javac or not java) and home:     # a JDK that is installed but not on PATH
        cand = [os.path.join(home, "bin", x) for x in ("javac", "java")]
        if all(os.access(x, os.X_OK) for x in cand):
            javac, java = cand
    if not javac or not java:
        return {"available": False, "reason": "no JDK on this box (javac/java neither on PATH nor under JAVA_HOME)"}
    try:
        out = tempfile.mkdtemp(prefix="lpx_java_")
        subprocess.check_call([javac, "-d", out, os.path.join(ROOT, "bench_java", "PivotBench.java")], timeout=120)
        ms, ns, piv = 1024, 2048, 10
        r = subprocess.run([java, "-Xmx8g", "-cp", out, "PivotBench", str(ms), str(ns), str(piv), "1"],
                           capture_output=True, text=True, timeout=max(60, 6 * budget_s))
        tok = [ln for ln in r.stdout.splitlines() if ln.startswith("JAVA_PIVOTS_PER_SEC")][-1].split()
        pps, done, secs = float(tok[1]), int(tok[2]), float(tok[3])
        scale = (float(m_full) * n) / (ms * ns)
        ver = subprocess.run([java, "-version"], capture_output=True, text=True).stderr.splitlines()[0]
        return {"available": True, "value": pps / scale, "unit": "pivots/s", "cores": 4, "kind": "reference-restatement",
                "sample": "%d BigDecimal pivots (MathContext(15,HALF_UP), 4 threads, pivotConcurrently's partitions) at "
                          "%dx%d in %.2f s, scaled x%.3g by element count; %s" % (done, ms, ns, secs, scale, ver)}
    except Exception as ex:  # never let the optional baseline break the bench line
        return {"available": False, "reason": "java harness failed: %r" % (ex,)}


def cpu_baselines(A, b, c, m_full, budget_s):
    """Times the CPU oracle on the first rows of the tableau (bounded sample) and scales to the full height:
    one pivot costs exactly m*n element updates, so time scales linearly in the number of rows."""
    from oracle import pyoracle as orc
    orc.build()
    out = {}
    m_s, n = A.shape
    scale = m_full / float(m_s)
    cores = host_cores()
    # fp64, all cores of this process's CPU share
    st = orc.State(A, b, c, kind=orc.FP64, with_perm=True)
    st.simplex_loop(max_pivots=1, threads=cores)  # touch pages / spin up the thread team
    piv = 8
    r = st.simplex_loop(max_pivots=piv, threads=cores)
    per = r["seconds"] / max(1, r["pivots"])
    out["cpu_baseline_fp64"] = {
        "value": 1.0 / (per * scale), "unit": "pivots/s", "cores": cores, "kind": "port",
        "sample": "%d pivots of the fp64 oracle (OpenMP row blocks) on rows 0..%d of the same tableau (%dx%d), "
                  "per-pivot time scaled x%.3g to m=%d" % (r["pivots"], m_s - 1, m_s, n, scale, m_full)}
    st.close()
    # decimal-15 (reference arithmetic), 4 threads = THREAD_AMOUNT of pivotConcurrently (LPState.java:22)
    rows_d = m_s
    est_per = 60e-9 * rows_d * n / 2.0  # ~60 ns per element update, ~2x from 4 threads
    while rows_d > 256 and est_per * 2 > budget_s:
        rows_d //= 2
        est_per /= 2
    Ad, bd = A[:rows_d], b[:rows_d]
    st = orc.State(Ad, bd, c, kind=orc.DEC15, with_perm=True)
    r = st.simplex_loop(max_pivots=2, threads=4)
    per = r["seconds"] / max(1, r["pivots"])
    scale_d = m_full / float(rows_d)
    out["cpu_baseline"] = {
        "value": 1.0 / (per * scale_d), "unit": "pivots/s", "cores": 4, "kind": "port",
        "sample": "%d pivots of the decimal-15 oracle (BigDecimal/MathContext(15,HALF_UP) semantics, 4 threads "
                  "with pivotConcurrently's static row partition) on rows 0..%d of the same tableau (%dx%d), "
                  "per-pivot time scaled x%.3g to m=%d" % (r["pivots"], rows_d - 1, rows_d, n, scale_d, m_full)}
    st.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default=os.environ.get("LPX_BENCH_WORKLOAD", "cfg4"), choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg3", action="store_true",
                    help="N=1, cfg4: skip the extra `cfg3` object (BASELINE.md quotes its single-GPU roofline target on "
                         "cfg3, so it is measured in the same run by default; a profile of ONE workload wants it off)")
    ap.add_argument("--no-steady", action="store_true",
                    help="N=1, cfg4: skip the `steady` object (the default loop for --steady-steps pivots after "
                         "--steady-warmup, cfg4 and cfg3, added when the headline run itself is shorter than that)")
    ap.add_argument("--no-fused", action="store_true",
                    help="N=1, cfg4: skip the `steady_plain` (`steady_fused`) object: the steady-state protocol on fresh handles "
                         "in the arithmetic mode the library did NOT choose by size, cfg4 and cfg3, replayed on the matching "
                         "oracle instantiation")
    ap.add_argument("--no-onepass", action="store_true",
                    help="skip the `onepass` object: the bandwidth-bound schedule north_star describes (one tableau pass "
                         "per pivot at N=1; two pivots per pass, the smallest block of the multi-GPU handle, at every N)")
    ap.add_argument("--onepass-steps", type=int, default=96)
    ap.add_argument("--onehop-steps", type=int, default=256)
    ap.add_argument("--steady-steps", type=int, default=512)
    ap.add_argument("--steady-warmup", type=int, default=64)
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the oracle replay that checks what the timed region computed")
    ap.add_argument("--parity-max-pivots", type=int, default=1600,
                    help="largest warm-up + steps the oracle replay is run for (cfg4: ~30 ms per pivot on 16 threads)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="lpx_state_set_option on the timed handle (names: linear_programming_solver_amd._lib.OPTIONS)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--poll-every", type=int, default=64,
                    help="sharded runs: pivots issued between two host polls of the replicated status word")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket every N-th row-update launch of the timed region with a HIP event pair "
                         "(an event pair costs ~3 us of stream time; 1 = every launch, 0 = none)")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="sharded runs: plain propose/all-gather/commit per pivot instead of the software-pipelined "
                         "form that overlaps the exchange of pivot t+1 with the row update of pivot t")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("LPX_BENCH_PIPELINE", "2")), choices=[1, 2],
                    help="sharded look-ahead form: 1 = in-place update, peek before it; 2 = out-of-place update "
                         "between two tableau buffers, peek + exchange + decision all beside the running update")
    ap.add_argument("--block", type=int, default=int(os.environ.get("LPX_BENCH_BLOCK", "-1")),
                    help="sharded runs: pivots per sweep (blocked pivoting); -1 = 16 when one rank's sweep is worth "
                         "several decisions, else 1 (then the look-ahead pipeline is used); 1 = off")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-block shard engine + the torch.distributed collective even at N=1")
    ap.add_argument("--multi-backend", default=os.environ.get("LPX_BENCH_MULTI", "peer"), choices=["peer", "rccl"],
                    help="N>1: 'peer' = lpx_multi through the C ABI (ONE process — rank 0 — drives all N GPUs, the "
                         "persistent decision kernels exchange candidates and pivot rows by direct xGMI stores; the "
                         "other ranks only join the barriers); 'rccl' = one process per GPU, one RCCL all_gather per "
                         "decision issued from Python (linear_programming_solver_amd/sharded.py)")
    ap.add_argument("--rehearse-shards", type=int, default=0,
                    help="N=1: run the lpx_multi path with this many shards, all on this GPU (peer-to-self rehearsal; "
                         "needs GPU_MAX_HW_QUEUES >= shards in the environment)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout: RCCL prints a version banner to fd 1 at init, so everything
    # else (C-level output included) is routed to stderr and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from linear_programming_solver_amd import LPState, _lib
    from linear_programming_solver_amd.sharded import DistExchange, HipShardEngine, row_block, sharded_simplex_loop

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed launcher (one process per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    peer = (world > 1 and args.multi_backend == "peer") or (world == 1 and args.rehearse_shards > 0)
    sharded = (world > 1 or args.force_sharded) and not peer
    if peer and world > 1:
        # one process drives every GPU: the other ranks only take part in the barriers (CPU backend, no GPU work)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        pg_opts = None
        try:  # RCCL's kernels run beside the streaming row update (look-ahead pipeline): high-priority stream
            pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            pg_opts = None
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank), pg_options=pg_opts)

    m, n = WORKLOADS[args.workload]
    nshards = world if world > 1 else max(1, args.rehearse_shards)
    if peer:
        r0, r1 = (0, m) if rank == 0 else (0, 0)     # rank 0 holds the whole problem and cuts it into row blocks
    else:
        r0, r1 = row_block(m, world, rank)
    t_gen = time.time()
    A, b, c = gen_rows(m, n, args.seed, r0, r1) if r1 > r0 else (None, None, None)
    t_gen = time.time() - t_gen
    K, W = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        if not (peer and rank != 0):
            torch.cuda.synchronize()

    options = {}
    for kv in args.option:
        k, v = kv.split("=", 1)
        options[k] = int(v)

    def single_path_extra(r):
        return isinstance(r, dict) and r.get("extra_sampling_calls", 0) > 0

    def is_fused(inf):
        """the arithmetic a handle really computes in (lpx_state_info.arith_fused: LPX_OPT_FUSED resolved — by size unless
        the --option set says otherwise), which picks the oracle instantiation that replays it"""
        if isinstance(inf, dict) and "arith_fused" in inf:
            return bool(inf["arith_fused"])
        return options.get("fused", 0) == 1

    def run_single(Aw, bw, cw, mw, nw, steps=None, warmup=None, st=None, done=0, opts=None):
        """warm-up + timed region of the single-GPU device loop on one workload; returns the measurements.
        st: continue on this handle (it has done `done` pivots) instead of uploading the tableau again.
        opts: the handle's options (default: the --option set)."""
        Ks = K if steps is None else steps
        Ws = W if warmup is None else warmup
        t_up = time.perf_counter()
        if st is None:
            st = LPState(Aw, bw, cw, device=local_rank, options=options if opts is None else opts)
        t_up = time.perf_counter() - t_up
        status, piv, _ = st.simplex_loop(max_pivots=Ws)
        assert piv == Ws, "LP finished during warm-up (status %d after %d pivots)" % (status, piv)
        block = st.block()
        # one-pass form: sample every N-th row update (an event pair costs ~3 us); blocked form: few, long sweeps: all
        st.profile_enable(args.event_every if block == 1 else (1 if args.event_every > 0 else 0))
        barrier()
        t0 = time.perf_counter()
        status, piv, _ = st.simplex_loop(max_pivots=Ks)
        barrier()
        elapsed = time.perf_counter() - t0
        launches, kernel_ms = st.profile_read()
        assert piv == Ks, "timed region did %d pivots instead of %d (status %d)" % (piv, Ks, status)
        extra_calls = 0
        if block > 1 and 0 < launches < 3 and args.event_every > 0:
            # the roofline figure of a short command (the driver's 20 steps are ONE sweep launch) rests on at least three
            # launches: the same budget again, outside the timed region, same kernel, same padding (Weak 8)
            while launches < 3 and extra_calls < 4:
                st.profile_enable(1)
                status, piv2, _ = st.simplex_loop(max_pivots=Ks)
                assert piv2 == Ks, "extra sampling call did %d pivots instead of %d (status %d)" % (piv2, Ks, status)
                l2, k2 = st.profile_read()
                launches, kernel_ms, extra_calls = launches + l2, kernel_ms + k2, extra_calls + 1
        st.profile_enable(False)
        if block > 1:
            # blocked pivoting: blocks of `block` decisions while the budget lasts, the tail (incl. the decision that
            # only reports the end of the budget) in one last block; a block made of that decision alone has no sweep
            expect = 0 if args.event_every <= 0 else (Ks + block) // block if (Ks % block) else Ks // block
            expect *= 1 + extra_calls
            pivots_per_launch = Ks * (1 + extra_calls) / float(launches) if launches else float("nan")
        else:
            expect = 0 if args.event_every <= 0 else (Ks + args.event_every - 1) // args.event_every
            pivots_per_launch = 1.0
        assert launches == expect, "sampled %d row-update launches, expected %d" % (launches, expect)
        avg_ms = kernel_ms / launches if launches else float("nan")
        return {"st": st, "elapsed": elapsed, "block": block, "launches": launches, "avg_ms": avg_ms, "steps": Ks,
                "warmup": Ws, "done": done + Ws + Ks * (1 + extra_calls), "pivots_per_launch": pivots_per_launch, "upload_s": t_up,
                "extra_sampling_calls": extra_calls,
                "info": st.info(), "fused": bool(st.info().get("arith_fused", 0))}

    def measured(r, mw, nw, name, Aw, bw, cw, with_parity):
        """one measurement as an object of the JSON line: value, roofline of its sweep / row-update launch, oracle replay"""
        kern = kernel_label(r["block"], r["info"])
        traffic, tsrc = load_traffic(name, 1, r["block"], kern, fused=r.get("fused", False))
        o = {"workload": "%s: m=%d n=%d, %d pivots after %d warm-up" % (name, mw, nw, r["steps"], r["warmup"]),
             "value": r["steps"] / r["elapsed"], "unit": "pivots/s", "ms_per_step": 1e3 * r["elapsed"] / r["steps"],
             "steps": r["steps"], "warmup": r["warmup"], "pivots_per_sweep": r["block"],
             "roofline": roofline_block(mw, nw, r["pivots_per_launch"], r["avg_ms"], kern, r["launches"], traffic, tsrc,
                                        fused=r.get("fused", False), clock_mhz=(r.get("info") or {}).get("sweep_clock_mhz", 0),
                                        cus=(r.get("info") or {}).get("sweep_cus", 0)),
             "loop_bound": loop_bound(1e3 * r["elapsed"] / r["steps"], r["pivots_per_launch"], r["avg_ms"], r.get("info"))}
        if with_parity:
            o["parity_after_timed_region"] = parity_after(r["st"], Aw, bw, cw, r["done"], mw, nw, host_cores(),
                                                          args.parity_max_pivots, fused=r.get("fused", False))
        return o

    def peer_selfcheck(devices):
        """Before anything is timed on a device set that has never been exercised together: a small LP through the
        multi-device handle must leave exactly the bits the one-device handle leaves (70 pivots: two blocks + a tail,
        the owner of the leaving row changing from decision to decision)."""
        from linear_programming_solver_amd import LPMulti
        rng = np.random.default_rng(7)
        ms, ns = 64 * len(devices), 1024
        As, bs, cs = rng.random((ms, ns)), (ns / 4.0) * (1.0 + rng.random(ms)), rng.random(ns)
        one = LPState(As, bs, cs, device=devices[0], block=16)
        many = LPMulti(As, bs, cs, devices=devices, block=16)
        try:
            r1 = one.simplex_loop(max_pivots=70)
            r2 = many.simplex_loop(max_pivots=70)
            g1, g2 = one.read(), many.read()
            same = r1[:2] == r2[:2] and all(np.array_equal(np.asarray(x).view(np.uint64), np.asarray(y).view(np.uint64))
                                            for x, y in zip(g1[:3], g2[:3])) and g1[3] == g2[3] and list(g1[4]) == list(g2[4])
            if not same:
                raise RuntimeError("lpx_multi self-check: the sharded result differs from the one-device result")
        finally:
            one.close()
            many.close()

    def run_peer(Aw, bw, cw):
        """the same protocol on the lpx_multi handle: rank 0 drives all shards, every rank joins the barriers.  Any
        failure on rank 0 (the peer path has not run on real multi-GPU hardware before) is reported to all ranks,
        which then fall back to the RCCL form."""
        from linear_programming_solver_amd import LPMulti
        mt, t_up, block, launches, avg_ms, ppl, info = None, 0.0, 0, 0, float("nan"), float("nan"), None
        err = None
        if rank == 0:
            try:
                devices = list(range(world)) if world > 1 else [local_rank] * nshards
                if world > 1:
                    peer_selfcheck(devices)
                t_up = time.perf_counter()
                mt = LPMulti(Aw, bw, cw, devices=devices, options=options)
                t_up = time.perf_counter() - t_up
                status, piv, _ = mt.simplex_loop(max_pivots=W)
                assert piv == W, "LP finished during warm-up (status %d after %d pivots)" % (status, piv)
                mt.profile_enable(1 if args.event_every > 0 else 0)
            except Exception as ex:   # noqa: BLE001 - reported, then the RCCL form is used
                err = "%s: %s" % (type(ex).__name__, ex)
        barrier()
        t0 = time.perf_counter()
        if rank == 0 and err is None:
            try:
                status, piv, _ = mt.simplex_loop(max_pivots=K)
                assert piv == K, "timed region did %d pivots instead of %d (status %d)" % (piv, K, status)
            except Exception as ex:   # noqa: BLE001
                err = "%s: %s" % (type(ex).__name__, ex)
        barrier()
        elapsed = time.perf_counter() - t0
        onepass, onehop, grid_leg, ref_1gpu = None, None, None, None
        counted = [0]               # pivots done behind the timed region, counted as they are done (the replay covers them)

        def extra_loop(handle, budget):
            piv = handle.simplex_loop(max_pivots=budget)[1]
            if handle is mt:
                counted[0] += int(piv)
            return int(piv)

        if rank == 0 and err is None and not args.no_onepass:
            # the bandwidth-bound leg: blocks of 2 pivots (the smallest the multi-GPU handle runs), so that a scaling
            # record shows what row sharding multiplies (the sweep) next to the blocked number it barely changes
            try:
                mt.set_option("block", 2)
                extra_loop(mt, 16)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                piv2 = extra_loop(mt, args.onepass_steps)
                dt = time.perf_counter() - t1
                onepass = {"pivots_per_pass": 2, "steps": int(piv2), "value": piv2 / dt, "unit": "pivots/s",
                           "ms_per_step": 1e3 * dt / max(1, piv2),
                           "what": "lpx_multi with blocks of 2 pivots: per pivot half a pass over every shard's rows "
                                   "(16*m*n/2 bytes in all) + one decision"}
            except Exception as ex:   # noqa: BLE001 - the extra leg never breaks the line
                onepass = {"error": "%s: %s" % (type(ex).__name__, ex)}
            finally:
                mt.set_option("block", options.get("block", 0))
        if rank == 0 and err is None and nshards > 1 and not args.no_onepass:
            # the same blocked loop with the one-hop exchange (every shard ships its candidate's row with its candidate:
            # one cross-device hop per decision instead of two); bit-identical, so the replay below covers it as well
            try:
                mt.set_option("multi_onehop", 1)
                extra_loop(mt, 32)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                piv3 = extra_loop(mt, args.onehop_steps)
                dt = time.perf_counter() - t1
                onehop = {"steps": int(piv3), "value": piv3 / dt, "unit": "pivots/s", "ms_per_step": 1e3 * dt / max(1, piv3),
                          "used": int(mt.info().get("multi_onehop", 0)),
                          "what": "the blocked loop with LPX_OPT_MULTI_ONEHOP = 1 (opt-in; `value` is the default two-hop form)"}
            except Exception as ex:   # noqa: BLE001
                onehop = {"error": "%s: %s" % (type(ex).__name__, ex)}
            finally:
                mt.set_option("multi_onehop", options.get("multi_onehop", 0))
        if rank == 0 and err is None and world > 1 and not args.no_onepass:
            # third leg of the A/B (real devices only: with all shards on one GPU two 65-workgroup grids are not resident): one-hop exchange AND the by-size decision grid of the one-device loop (one row / one
            # column per thread on 8 CUs per XCD), on a handle of its own (the CU share is fixed when a handle's streams
            # are made); replayed on the oracle by itself
            m2 = None
            try:
                m2 = LPMulti(Aw, bw, cw, devices=devices, options=dict(options, multi_onehop=1, chain_wgs=65, chain_cus=8))
                extra_loop(m2, 64)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                piv4 = extra_loop(m2, args.onehop_steps)
                dt = time.perf_counter() - t1
                grid_leg = {"steps": int(piv4), "value": piv4 / dt, "unit": "pivots/s", "ms_per_step": 1e3 * dt / max(1, piv4),
                            "engine": m2.info(),
                            "what": "one-hop exchange + decision grid chain_wgs = 65 on chain_cus = 8 CUs per XCD (opt-in)"}
                if not args.no_parity:
                    grid_leg["parity_after_timed_region"] = parity_after(m2, Aw, bw, cw, 64 + int(piv4), m, n, host_cores(),
                                                                         args.parity_max_pivots,
                                                                         fused=is_fused(m2.info()))
            except Exception as ex:   # noqa: BLE001
                grid_leg = {"error": "%s: %s" % (type(ex).__name__, ex)}
            finally:
                if m2 is not None:
                    m2.close()
        if rank == 0 and err is None and world > 1 and not args.no_onepass:
            # what the N-GPU numbers are speed-ups OF: the same job on GPU 0 alone, blocked (the default loop) and one
            # pass per pivot (the schedule north_star's >= 6x was written against), measured in this very run
            s1 = None
            try:
                s1 = LPState(Aw, bw, cw, device=devices[0], options=options)
                s1.simplex_loop(max_pivots=W)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                p1 = s1.simplex_loop(max_pivots=K)[1]
                dt1 = time.perf_counter() - t1
                s1.set_option("block", 1)
                s1.simplex_loop(max_pivots=8)
                t1 = time.perf_counter()
                p2 = s1.simplex_loop(max_pivots=max(8, args.onepass_steps // 2))[1]
                dt2 = time.perf_counter() - t1
                ref_1gpu = {"blocked": p1 / dt1, "one_pass_per_pivot": p2 / dt2, "unit": "pivots/s",
                            "arith_fused": int(s1.info().get("arith_fused", 0)),
                            "what": "the same tableau on GPU %d alone in this run: %d pivots after %d warm-up (default loop: the "
                                    "one-device handle chooses its arithmetic by size — arith_fused — while shards keep the two "
                                    "roundings per update unless --option fused=1), then %d pivots with one pass per pivot"
                                    % (devices[0], p1, W, p2)}
            except Exception as ex:   # noqa: BLE001
                ref_1gpu = {"error": "%s: %s" % (type(ex).__name__, ex)}
            finally:
                if s1 is not None:
                    s1.close()
        extra_pivots = counted[0]
        barrier()
        if dist is not None:
            box = [err]
            dist.broadcast_object_list(box, src=0)
            err = box[0]
        if err is not None:
            if mt is not None:
                mt.close()
            return {"error": err}
        if rank == 0:
            launches, kernel_ms = mt.profile_read(0)
            mt.profile_enable(0)
            info = mt.info()
            block = info["block"]
            avg_ms = kernel_ms / launches if launches else float("nan")
            ppl = K / float(launches) if launches else float("nan")
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return {"st": mt, "elapsed": elapsed, "block": block, "launches": launches, "avg_ms": avg_ms,
                "pivots_per_launch": ppl, "upload_s": t_up, "info": info, "onepass": onepass, "onehop": onehop,
                "grid_leg": grid_leg, "ref_1gpu": ref_1gpu, "pivots_done": W + K + extra_pivots}

    fallback_reason = None
    if peer:
        r1_ = run_peer(A, b, c)
        if "error" in r1_:
            if world == 1:
                raise SystemExit("lpx_multi rehearsal failed: " + r1_["error"])
            # the documented alternative: one process per GPU, one RCCL all_gather per decision
            fallback_reason = r1_["error"]
            sys.stderr.write("bench.py: lpx_multi path failed (%s); falling back to --multi-backend rccl\n" % fallback_reason)
            peer, sharded = False, True
            dist.barrier()
            dist.destroy_process_group()
            pg_opts = None
            try:
                pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                pg_opts = None
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank), pg_options=pg_opts)
            r0, r1 = row_block(m, world, rank)
            A, b, c = gen_rows(m, n, args.seed, r0, r1)
    if peer:
        eng = st = r1_["st"]
        elapsed, block, launches, avg_ms = r1_["elapsed"], r1_["block"], r1_["launches"], r1_["avg_ms"]
        pivots_per_launch, t_up, info = r1_["pivots_per_launch"], r1_["upload_s"], r1_["info"]
        objective = st.v if rank == 0 else None
    elif not sharded:
        r1_ = run_single(A, b, c, m, n)
        eng = st = r1_["st"]
        elapsed, block, launches, avg_ms = r1_["elapsed"], r1_["block"], r1_["launches"], r1_["avg_ms"]
        pivots_per_launch, t_up, info = r1_["pivots_per_launch"], r1_["upload_s"], r1_["info"]
        objective = st.v
    else:
        t_up = time.perf_counter()
        block = args.block
        if block < 0:   # one decision incl. the exchange ~60 us; a rank's sweep 16*m_local*n bytes at ~6 TB/s
            sweep_us = 16.0 * (r1 - r0) * n / 6.0e6
            block = 1 if sweep_us <= 50.0 else (16 if sweep_us < 250.0 else 32)
        # blocked form: nothing runs beside the sweep, so it keeps all 8 XCDs (the look-ahead pipeline reserves one
        # for the exchange and the decision kernels)
        eng = HipShardEngine(A, b, c, r0, m, world, device=local_rank, pipeline=args.pipeline if block == 1 else 1,
                             reserve_xcds=1 if block == 1 else 0)
        t_up = time.perf_counter() - t_up
        ex = DistExchange()
        status, piv, _ = sharded_simplex_loop([eng], ex, max_pivots=W, poll_every=args.poll_every,
                                              lookahead=not args.no_lookahead, block=block)
        assert piv == W, "LP finished during warm-up (status %d after %d pivots)" % (status, piv)
        eng.profile_enable(args.event_every if block == 1 else (1 if args.event_every > 0 else 0))
        barrier()
        t0 = time.perf_counter()
        status, piv, _ = sharded_simplex_loop([eng], ex, max_pivots=K, poll_every=args.poll_every,
                                              lookahead=not args.no_lookahead, block=block)
        barrier()
        elapsed = time.perf_counter() - t0
        launches, kernel_ms = eng.profile_read()
        eng.profile_enable(False)
        objective = eng.read(want_A=False)[3]
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert piv == K, "timed region did %d pivots instead of %d (status %d)" % (piv, K, status)
        info = None
        if block > 1:   # ceil((K+1)/block) sweeps (the last decision of a budgeted run only reports the end)
            expect = 0 if args.event_every <= 0 else (K + 1 + block - 1) // block
            pivots_per_launch = K / float(launches) if launches else float("nan")
        else:
            block = 1
            expect = 0 if args.event_every <= 0 else (K + args.event_every - 1) // args.event_every
            pivots_per_launch = 1.0
        assert launches == expect, "sampled %d row-update launches, expected %d" % (launches, expect)
        avg_ms = kernel_ms / launches if launches else float("nan")

    if rank == 0:
        m_local = (m // nshards) if peer else (r1 - r0)      # rows one sweep launch covers (shard 0)
        line = {
            "metric": "simplex_pivots_per_sec", "value": K / elapsed, "unit": "pivots/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: dense random LP m=%d n=%d fp64 (A~U(0,1), b=(n/4)U(1,2), c~U(0,1), max), "
                                   "first-positive entering rule, %d pivots after %d warm-up" % (args.workload, m, n, K, W),
                       "m": m, "n": n, "seed": args.seed, "pivots_per_sweep": block,
                       "parallelism": (
                           ("row-block x%d behind one lpx_multi handle (one process, %s): per decision a 32-byte "
                            "candidate to every peer + the pivot row from its owner by direct stores, blocked x%d" % (
                                nshards, "%d GPUs, xGMI peer access" % world if world > 1 else
                                "REHEARSAL: all shards on this one GPU", block)) if peer else
                           "single GPU" if not sharded else "row-block x%d, 1 all_gather/pivot, %s" % (
                               world, ("blocked x%d" % block) if block > 1 else
                               ("plain" if args.no_lookahead else "look-ahead pipeline %d" % args.pipeline)))},
            "roofline": roofline_block(m_local, n, pivots_per_launch, avg_ms, kernel_label(block, info), launches,
                                       *load_traffic(args.workload, world, block, kernel_label(block, info), fused=is_fused(info)),
                                       fused=is_fused(info), clock_mhz=(info or {}).get("sweep_clock_mhz", 0),
                                       cus=(info or {}).get("sweep_cus", 0)),
            "loop_bound": loop_bound(1e3 * elapsed / K, pivots_per_launch, avg_ms, info),
            "objective_after_timed_region": objective,
            "host_gen_s": t_gen,
            "host_upload_s": t_up,   # hipMalloc + PCIe upload of this rank's tableau; outside the timed region
        }
        if single_path_extra(r1_):
            line["roofline"]["sampled_outside_timed_region"] = (
                "%d further calls of the same %d-pivot budget behind the timed region (same kernel, same padding): the figure "
                "rests on %d launches instead of one" % (r1_["extra_sampling_calls"], K, launches))
        if info is not None:   # what the engine actually did: grid of the decision kernel, its residency bound, CU masks
            line["engine"] = info
        if world > 1:
            line["multi_backend"] = "peer (lpx_multi)" if peer else ("rccl" if fallback_reason is None else
                                                                     "rccl, after the lpx_multi path failed: " + fallback_reason)
        if (peer or (world == 1 and not sharded)) and not args.no_parity:
            # the checker: the same LP replayed on the fp64 oracle for warm-up + steps pivots (outside the timed region)
            # (on the multi-GPU handle the bandwidth-bound leg has run behind the timed region: the replay covers it too)
            # (a short command's extra sampling calls — roofline.launches_sampled >= 3 — have run on the handle too)
            line["parity_after_timed_region"] = parity_after(st, A, b, c, r1_.get("pivots_done", W + K) if peer else r1_.get("done", W + K),
                                                             m, n, host_cores(), args.parity_max_pivots,
                                                             fused=is_fused(info))
        line["devices_visible"] = torch.cuda.device_count()
        if peer and r1_.get("onepass") is not None:
            line["onepass"] = r1_["onepass"]
        if peer and r1_.get("onehop") is not None:
            line["onehop"] = r1_["onehop"]
        if peer and r1_.get("grid_leg") is not None:
            line["onehop_grid"] = r1_["grid_leg"]
        if peer and isinstance(r1_.get("ref_1gpu"), dict):
            ref = r1_["ref_1gpu"]
            line["ref_1gpu"] = ref
            if "error" not in ref:
                # side by side: north_star's >= 6x was written against the one-pass schedule
                line["speedup_vs_blocked_1gpu"] = line["value"] / ref["blocked"]
                if isinstance(r1_.get("onepass"), dict) and "error" not in r1_["onepass"]:
                    line["speedup_onepass_vs_onepass_1gpu"] = r1_["onepass"]["value"] / ref["one_pass_per_pivot"]
                line["speedup_vs_onepass_1gpu"] = line["value"] / ref["one_pass_per_pivot"]
        single = world == 1 and not sharded and not peer
        steady = {}
        want_steady = single and args.workload == "cfg4" and not args.no_steady

        def steady_leg(r_first, Aw, bw, cw, mw, nw, name, with_parity):
            """The steady state of the default loop (>= 512 pivots after >= 64 warm-up), driver-run: the headline of a
            short command (the driver's 20 steps: one decision launch + one sweep, nothing overlapped, clocks still
            ramping) says little about the loop a solve spends its time in.  Continues on the same handle."""
            if r_first["steps"] >= args.steady_steps and r_first["warmup"] >= args.steady_warmup:
                return None, r_first["done"]   # the headline IS a steady-state measurement
            rs = run_single(None, None, None, mw, nw, steps=args.steady_steps, warmup=args.steady_warmup,
                            st=r_first["st"], done=r_first["done"])
            return measured(rs, mw, nw, name, Aw, bw, cw, with_parity), rs["done"]

        default_fused = is_fused(info)   # what the library chose for this workload (LPX_OPT_FUSED = 2: by size)

        def fused_leg(Aw, bw, cw, mw, nw, name, block=None):
            """The same steady-state protocol on a FRESH handle in the OTHER arithmetic mode than the one the headline ran
            in: with the by-size default (fused multiply-add updates from 0.5 GiB: one v_fma_f64 per update — or, in blocks
            of 33..64, four of them per v_mfma_f64_16x16x4) that is the opt-out, LPX_OPT_FUSED = 0 (product and difference
            rounded separately); replayed on the matching oracle instantiation.  block: pivots per sweep (None: by size)."""
            fo = dict(options, fused=0 if default_fused else 1)
            if block is not None:
                fo["block"] = block
            rf = run_single(Aw, bw, cw, mw, nw, steps=args.steady_steps, warmup=args.steady_warmup, opts=fo)
            try:
                return measured(rf, mw, nw, name, Aw, bw, cw, not args.no_parity)
            finally:
                rf["st"].close()

        want_fused = want_steady and not args.no_fused and "fused" not in options
        steady_fused = {}
        base_done = r1_.get("done", W + K) if single else W + K   # pivots the cfg4 handle has done (every leg below continues on it)
        done_total = base_done
        if want_steady:
            # (its oracle replay comes after the one-pass legs, which continue on the same handle: one replay of everything)
            leg, done_total = steady_leg(r1_, A, b, c, m, n, "cfg4", False)
            if leg is not None:
                steady["cfg4"] = leg
        if single and not args.no_onepass:
            # the schedule north_star describes, on the same handle: one pass per pivot (k_update: the 16*m*n-bytes-per-
            # pivot roofline kernel), and two pivots per pass as the multi-GPU handle's bandwidth-bound leg runs them
            counted = [done_total]

            def one_leg(block_opt, steps):
                st.set_option("block", block_opt)
                counted[0] += st.simplex_loop(max_pivots=8)[1]
                st.profile_enable(1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, piv2, _ = st.simplex_loop(max_pivots=steps)
                counted[0] += piv2
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                ln, kms = st.profile_read()
                st.profile_enable(False)
                inf = st.info()
                return {"pivots_per_pass": block_opt, "steps": int(piv2), "value": piv2 / dt, "unit": "pivots/s",
                        "ms_per_step": 1e3 * dt / max(1, piv2),
                        "roofline": roofline_block(m, n, piv2 / float(ln) if ln else float("nan"),
                                                   kms / ln if ln else float("nan"), kernel_label(block_opt, inf), ln,
                                                   fused=is_fused(inf), cus=inf.get("sweep_cus", 0))}
            try:
                line["onepass"] = dict(one_leg(2, args.onepass_steps), one_pass_per_pivot=one_leg(1, args.onepass_steps // 2))
            except Exception as ex:   # noqa: BLE001 - the extra leg never breaks the line
                line["onepass"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            finally:
                st.set_option("block", options.get("block", 0))
                done_total = counted[0]
        if single and not args.no_parity and done_total > base_done:
            # ONE replay of everything the cfg4 handle has done: headline, steady leg, one-pass legs (k_update_tiles<2> and
            # k_update at full height are checked here too)
            final = parity_after(st, A, b, c, done_total, m, n, host_cores(), args.parity_max_pivots,
                                 fused=is_fused(info))
            final["covers"] = "warm-up + steps, the steady leg and the one-pass legs: every pivot this handle has done"
            if "cfg4" in steady:
                steady["cfg4"]["parity_after_timed_region"] = final
            if isinstance(line.get("onepass"), dict) and "error" not in line["onepass"]:
                line["onepass"]["parity_after_timed_region"] = final
        if single and args.workload == "cfg4" and (want_fused or not args.no_cfg3):
            st.close()
        if want_fused:
            try:
                steady_fused["cfg4"] = fused_leg(A, b, c, m, n, "cfg4")
            except Exception as ex:   # noqa: BLE001 - the extra leg never breaks the line
                steady_fused["cfg4"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if single and args.workload == "cfg4" and not args.no_cfg3:
            # BASELINE.md quotes its single-GPU roofline target on cfg3 (m=8192, n=16384): measure it in the same
            # run, same protocol, as an extra object (the headline `value` above stays the cfg4 job)
            m3, n3 = WORKLOADS["cfg3"]
            A3, b3, c3 = gen_rows(m3, n3, args.seed, 0, m3)
            r3 = run_single(A3, b3, c3, m3, n3)
            line["cfg3"] = measured(r3, m3, n3, "cfg3", A3, b3, c3, not args.no_parity)
            if want_steady:
                leg, _ = steady_leg(r3, A3, b3, c3, m3, n3, "cfg3", not args.no_parity)
                if leg is not None:
                    steady["cfg3"] = leg
            r3["st"].close()
            if want_fused:
                try:
                    steady_fused["cfg3"] = fused_leg(A3, b3, c3, m3, n3, "cfg3")
                except Exception as ex:   # noqa: BLE001
                    steady_fused["cfg3"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            del A3, b3, c3
        if steady:
            steady["protocol"] = ("the default loop continued on the same handle: %d more warm-up pivots, then %d timed "
                                  "pivots; roofline of its sweep launches; oracle replay of ALL pivots the handle has done"
                                  % (args.steady_warmup, args.steady_steps))
            line["steady"] = steady
        if steady_fused:
            steady_fused["protocol"] = ("a fresh handle in the other arithmetic mode (option fused = %d; `value` and `steady` "
                                        "run in the library's by-size choice, lpx_state_info.arith_fused = %d): %d warm-up "
                                        "pivots, then %d timed pivots; replayed on the matching oracle instantiation"
                                        % (0 if default_fused else 1, int(default_fused), args.steady_warmup, args.steady_steps))
            line["steady_plain" if default_fused else "steady_fused"] = steady_fused
        if world == 1 and not peer and not args.no_cpu_baseline:
            rows_s = min(m, 8192)
            line.update(cpu_baselines(A[:rows_s], b[:rows_s], c, m, args.cpu_budget_s))
            line["cpu_baseline_java"] = java_baseline(m, n, args.cpu_budget_s)
            line["gpu_over_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
            line["gpu_over_cpu_baseline_fp64"] = line["value"] / line["cpu_baseline_fp64"]["value"]
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if eng is not None and hasattr(eng, "close"):
        eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
