"""The hand-pipelined sweep kernels (k_sweep32_steady, k_sweep64_pipe) issue their tableau loads with `asm volatile`
and wait for them with an explicit `s_waitcnt vmcnt(N)` that names the destination registers.  That is only valid
while the compiler leaves those registers alone between the two — a spill of an in-flight destination would save and
restore garbage.  So the build must keep these kernels free of scratch (private segment size 0) and inside the
register budget of their occupancy; this test compiles the device code to assembly (no GPU needed) and checks the
resource lines the assembler prints for every instantiation."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "linear_programming_solver_amd", "csrc", "lpx_kernels.hip")
HIPCC = "/opt/rocm/bin/hipcc"


def _compile_asm(tmp_path_factory, fused):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm%d" % fused) / "lpx_kernels.s"
    # the flags of csrc/Makefile that matter for code generation (the file is compiled twice: LPX_FUSED = 0 / 1)
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-ffp-contract=off", "-Wno-unused-result", "-DLPX_FUSED=%d" % fused, "-S", "--cuda-device-only",
                           SRC, "-o", str(out)], stderr=subprocess.DEVNULL)
    return out.read_text()


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """lpxk::plain — the default arithmetic (product and difference rounded separately)."""
    return _compile_asm(tmp_path_factory, 0)


@pytest.fixture(scope="module")
def device_asm_fused(tmp_path_factory):
    """lpxk::fused — the opt-in fused-arithmetic compilation of the same file."""
    return _compile_asm(tmp_path_factory, 1)


def _resources(asm, kernel):
    """{mangled name: {num_vgpr, num_agpr, private_seg_size}} for every instantiation of `kernel`."""
    res = {}
    for name, key, val in re.findall(r"\.set (\S*%s\S*)\.(num_vgpr|num_agpr|private_seg_size), (\d+)" % kernel, asm):
        res.setdefault(name, {})[key] = int(val)
    return res


@pytest.mark.parametrize("kernel,max_vgpr", [("k_sweep32_steady", 256), ("k_sweep64_pipe", 256)])
def test_hand_pipelined_sweep_kernels_have_no_scratch(device_asm, kernel, max_vgpr):
    res = _resources(device_asm, kernel)
    assert len(res) == 4, sorted(res)   # <NT, OOP> x 2 x 2
    for name, r in res.items():
        assert r["private_seg_size"] == 0, (name, r)
        assert r["num_agpr"] == 0 and r["num_vgpr"] <= max_vgpr, (name, r)   # two waves per SIMD, nothing parked in AGPRs


def test_default_decision_kernel_has_no_scratch(device_asm):
    """The 32-slot decision kernel (the default loop) keeps its register arrays in registers: no scratch traffic on
    its latency-bound path.  (The opt-in 64-slot form spills a little; it is not checked here.)"""
    res = {k: v for k, v in _resources(device_asm, "k_block_chain_t").items() if "Li32E" in k}
    assert len(res) == 2, sorted(res)   # one device / shards of an lpx_multi
    for name, r in res.items():
        assert r["private_seg_size"] == 0, (name, r)


def _kernel_bodies(asm, kernel):
    """{mangled name: [instruction lines]} of every instantiation of `kernel` (labels and comments kept)."""
    out = {}
    for m in re.finditer(r"^(\S*%s\S*):\s*(?:;.*)?$" % kernel, asm, flags=re.M):
        end = asm.find("s_endpgm", m.end())
        out[m.group(1)] = [ln.strip() for ln in asm[m.end():end].splitlines()]
    return out


@pytest.mark.parametrize("kernel", ["k_sweep32_steady", "k_sweep64_pipe"])
def test_no_register_copy_sits_in_front_of_a_hand_written_wait(device_asm, kernel):
    """The hand-written `s_waitcnt vmcnt(N)` must come before ANY copy of the registers its loads are landing in.
    (A single asm with the registers as in/out operands let the allocator copy them in front of the statement: one
    wave's last batches of a run then came out stale once in ~15 000 workgroup runs.)  Within a basic block, between
    the previous asm statement (or the block's label) and a hand-written wait there must be no vector move."""
    bodies = _kernel_bodies(device_asm, kernel)
    assert len(bodies) == 4, sorted(bodies)
    for name, lines in bodies.items():
        waits = 0
        since = []          # instructions since the last label / asm statement
        in_asm = False
        for ln in lines:
            if ln.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if ln.startswith(";;#ASMEND"):
                in_asm = False
                since = []
                continue
            if in_asm:
                if ln.startswith("s_waitcnt vmcnt"):
                    waits += 1
                    moves = [x for x in since if x.startswith(("v_mov", "v_accvgpr", "scratch_"))]
                    assert not moves, (name, ln, moves)
                continue
            if ln.endswith(":") or ln.startswith(".LBB"):
                since = []
            elif ln and not ln.startswith(";"):
                since.append(ln)
        assert waits >= 3, (name, waits)


def _regs(operand):
    """VGPR numbers named by an operand such as v12 or v[34:37] (empty for anything else)."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("kernel", ["k_sweep32_steady", "k_sweep64_pipe"])
def test_no_register_copy_reads_a_hand_issued_load_before_its_wait(device_asm, kernel):
    """The other half of the same hazard: after a hand-issued `global_load` nothing may read its destination
    registers before a hand-written wait has been passed.  Checked in program order (the kernels' loops keep loads
    and the wait that covers them in straight succession): between an asm load and the next asm `s_waitcnt vmcnt`
    no move, scratch store or AGPR write may name a register the load is landing in."""
    bodies = _kernel_bodies(device_asm, kernel)
    for name, lines in bodies.items():
        landing = set()
        in_asm = False
        loads = 0
        for ln in lines:
            if ln.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if ln.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm:
                if ln.startswith("global_load_dword"):
                    landing |= _regs(ln.split()[1].rstrip(","))
                    loads += 1
                elif ln.startswith("s_waitcnt vmcnt(0)"):
                    landing = set()       # (a partial wait covers the older loads only: keep the set)
                continue
            if ln.startswith(("v_mov", "v_accvgpr_write", "scratch_store")):
                ops = [o.strip() for o in ln.split(None, 1)[1].split(",")]
                read = set().union(*[_regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                if ln.startswith("scratch_store"):
                    read = set().union(*[_regs(o) for o in ops])
                assert not (read & landing), (name, ln, sorted(read & landing))
        assert loads >= 8, (name, loads)


# ---- round 3: the LDS-DMA sweep kernels -------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["k_sweep32_dma", "k_sweep32_pull", "k_sweep64_pull", "k_sweep64_one"])
def test_lds_dma_sweep_kernels_have_no_scratch(device_asm, kernel):
    """Their tableau loads land in LDS, not in registers, but the pivot-row slices (128 VGPRs) must stay in registers
    and two workgroups must fit a CU: no scratch, no AGPRs, at most 256 VGPRs, LDS <= 80 KiB."""
    res = _resources(device_asm, kernel)
    assert len(res) == 4, sorted(res)
    for name, r in res.items():
        assert r["private_seg_size"] == 0 and r["num_agpr"] == 0 and r["num_vgpr"] <= 256, (name, r)
    lds = re.findall(r"\.amdhsa_group_segment_fixed_size (\d+)", "".join(
        device_asm[m.start():m.start() + 4000] for m in re.finditer(r"\.amdhsa_kernel \S*%s" % kernel, device_asm)))
    assert len(lds) == 4 and all(int(x) <= 80 * 1024 for x in lds), lds


@pytest.mark.parametrize("kernel", ["k_sweep32_pull", "k_sweep64_pull", "k_sweep64_one"])
def test_pulled_tickets_are_not_touched_before_they_are_taken(device_asm, kernel):
    """The pull kernels' only hand-issued operation with a register destination is the ticket atomic.  Between the asm
    statement that issues it and the v_readfirstlane that takes the ticket (behind a hand-written s_waitcnt vmcnt and
    a scheduling barrier) no compiler-generated instruction may name that register — a copy made earlier would carry
    the value from before the atomic returned (the hazard class of DESIGN.md 3a)."""
    bodies = _kernel_bodies(device_asm, kernel)
    assert len(bodies) == 4, sorted(bodies)
    for name, lines in bodies.items():
        labels = {ln[:-1]: k for k, ln in enumerate(lines) if ln.endswith(":") and ln.startswith(".LBB")}
        back = [labels[t] for k, ln in enumerate(lines) for t in re.findall(r"s_c?branch\S*\s+(\.LBB\S+)", ln)
                if t in labels and labels[t] < k]
        loop_start = min(back) if back else 0
        in_asm = [False] * len(lines)
        flag = False
        for k, ln in enumerate(lines):
            if ln.startswith(";;#ASMSTART"):
                flag = True
            in_asm[k] = flag
            if ln.startswith(";;#ASMEND"):
                flag = False
        atomics = [(k, ln.split()[1].rstrip(",")) for k, ln in enumerate(lines)
                   if in_asm[k] and ln.startswith("global_atomic_add")]
        assert len(atomics) >= 9, (name, atomics)   # prologue 2 x 3, loop 3
        for k, reg in atomics:
            order = list(range(k + 1, len(lines))) + list(range(loop_start, k))
            waited = False
            taken = False
            for j in order:
                ln = lines[j]
                if in_asm[j]:
                    waited = waited or ln.startswith("s_waitcnt vmcnt")
                    continue
                if not ln or ln.startswith(";") or ln.endswith(":"):
                    continue
                if re.search(r"\b%s\b" % reg, ln):
                    assert ln.startswith("v_readfirstlane_b32") and waited, (name, reg, ln, waited)
                    taken = True
                    break
            assert taken, (name, reg)


# ---- round 4 ---------------------------------------------------------------------------------------------------------
def test_round4_decision_kernel_keeps_its_arrays_in_registers(device_asm, device_asm_fused):
    """k_block_chain2_t<32, 256> (the default decision kernel of the one-device loop) in both compilations: one wave per
    SIMD, its register arrays in VGPRs / AGPRs.  At most a few dozen bytes of scratch (one 16-byte pair spilled around
    the rare full exchange at the end of a decision), nothing on the hand-off path: every scratch access sits behind the
    last poll of workgroup 0's record."""
    for asm in (device_asm, device_asm_fused):
        res = {k: v for k, v in _resources(asm, "k_block_chain2_t").items() if "Li32E" in k}
        assert len(res) == 1, sorted(res)
        for name, r in res.items():
            assert r["private_seg_size"] <= 64, (name, r)
        for name, lines in _kernel_bodies(asm, "k_block_chain2_t").items():
            if "Li32E" not in name:
                continue
            code = [ln for ln in lines if ln and not ln.startswith((";", "."))]
            scratch = [k for k, ln in enumerate(code) if ln.startswith("scratch_")]
            assert all(k > 0.85 * len(code) for k in scratch), (name, scratch, len(code))


def _sweep_arith(asm, kernel):
    """(v_fma_f64, v_mul_f64 + v_add_f64) instruction counts summed over the instantiations of a sweep kernel."""
    fma = unfused = 0
    for lines in _kernel_bodies(asm, kernel).values():
        fma += sum(1 for ln in lines if ln.startswith("v_fma_f64"))
        unfused += sum(1 for ln in lines if ln.startswith(("v_mul_f64", "v_add_f64")))
    return fma, unfused


@pytest.mark.parametrize("kernel", ["k_sweep32_pull", "k_sweep64_one", "k_update_tiles"])
def test_the_two_compilations_differ_in_the_update_arithmetic_only(device_asm, device_asm_fused, kernel):
    """lpxk::plain: every update is v_mul_f64 + v_add_f64 (two roundings, LPState.java:162) and no FMA is formed;
    lpxk::fused: one v_fma_f64 per update and no separate multiply / add — same count of updates in both."""
    pf, pu = _sweep_arith(device_asm, kernel)
    ff, fu = _sweep_arith(device_asm_fused, kernel)
    assert pf == 0 and pu > 0 and pu % 2 == 0, (kernel, pf, pu)
    assert fu == 0 and ff == pu // 2, (kernel, ff, fu, pu)


def test_mfma_sweep_fits_two_waves_per_simd(device_asm_fused):
    """k_sweep64_mfma2 (fused blocks of 33..64 on the matrix cores): 64 MFMAs per tile in three unrolled tile slots, no
    scratch, at most 256 registers in all (two waves per SIMD), 64 KiB of LDS for the B operands (two workgroups per CU);
    the default compilation has no such kernel."""
    res = _resources(device_asm_fused, "k_sweep64_mfma2")
    assert len(res) == 4, sorted(res)
    for name, r in res.items():
        assert r["private_seg_size"] == 0 and r["num_vgpr"] + r["num_agpr"] <= 256, (name, r)
    for name, lines in _kernel_bodies(device_asm_fused, "k_sweep64_mfma2").items():
        assert sum(1 for ln in lines if ln.startswith("v_mfma_f64_16x16x4")) == 192, name
    lds = re.findall(r"\.amdhsa_group_segment_fixed_size (\d+)", "".join(
        device_asm_fused[m.start():m.start() + 4000] for m in re.finditer(r"\.amdhsa_kernel \S*k_sweep64_mfma2", device_asm_fused)))
    assert len(lds) == 4 and all(int(x) <= 80 * 1024 for x in lds), lds


def test_mfma_sweep_loop_never_drains_its_memory_queue(device_asm_fused):
    """The memory side of k_sweep64_mfma2's loop as designed: buffer addressing (no 64-bit VALU address
    per load), a straight-line body with ONE backward branch, three hand-issued ticket atomics per round and, between
    the loop head and that branch, no vmcnt wait that could stall on anything younger than a step (the compiler's clamped
    counts, >= 62, and the exact wait in front of a ticket pulled a step earlier).  The first version had `vmcnt(0)` behind every
    aggregated atomic and, with exits between the steps, waits for the loads the previous step had just issued."""
    for name, lines in _kernel_bodies(device_asm_fused, "k_sweep64_mfma2").items():
        back = [k for k, ln in enumerate(lines) if re.match(r"s_cbranch_scc[01] \.LBB\d+_\d+", ln)]
        labels = {m.group(1): k for k, ln in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
        loops = [(labels[ln.split()[-1]], k) for k in back for ln in [lines[k]] if labels.get(ln.split()[-1], k) < k]
        assert loops, name
        head, tail = max(loops, key=lambda ht: ht[1] - ht[0])     # the loop with the longest body: the tile loop
        body = lines[head:tail]
        assert sum(1 for ln in body if ln.startswith("v_mfma_f64_16x16x4")) == 192, name
        assert sum(1 for ln in body if ln.startswith("global_atomic_add")) == 3, name
        assert not any(ln.startswith(("s_cbranch", "s_branch")) for ln in body), name
        assert sum(1 for ln in body if ln.startswith("buffer_load_dwordx2")) == 48, name      # three tiles of C
        assert sum(1 for ln in body if ln.startswith("buffer_load_dwordx4")) == 24, name      # three tiles of A operands
        assert sum(1 for ln in body if ln.startswith("buffer_store_dwordx2")) == 48, name
        assert not any(ln.startswith(("global_load", "global_store", "flat_", "v_lshl_add_u64")) for ln in body), name
        waits = [int(x) for ln in body for x in re.findall(r"vmcnt\((\d+)\)", ln)]
        # 41 = the hand-written wait in front of a ticket's take (one step's 16 + 8 loads, 16 stores and the next pull are
        # younger than the atomic); everything else is the compiler's, clamped
        assert waits.count(41) == 3 and all(w == 41 or w >= 62 for w in waits), (name, sorted(set(waits)))
