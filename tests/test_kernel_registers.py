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


def _compile_asm(tmp_path_factory, fused, variants=False):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm%d%d" % (fused, variants)) / "lpx_kernels.s"
    # the flags of csrc/Makefile that matter for code generation (the file is compiled twice: LPX_FUSED = 0 / 1; the
    # variants library adds -DLPX_WITH_VARIANTS: the superseded kernels of csrc/variants/)
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-ffp-contract=off", "-Wno-unused-result", "-DLPX_FUSED=%d" % fused] +
                          (["-DLPX_WITH_VARIANTS"] if variants else []) +
                          ["-S", "--cuda-device-only", SRC, "-o", str(out)], stderr=subprocess.DEVNULL)
    return out.read_text()


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """lpxk::plain — the default arithmetic (product and difference rounded separately)."""
    return _compile_asm(tmp_path_factory, 0)


@pytest.fixture(scope="module")
def device_asm_variants(tmp_path_factory):
    """lpxk::plain of the variants library (`make variants`): the product's kernels plus csrc/variants/."""
    return _compile_asm(tmp_path_factory, 0, variants=True)


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
def test_hand_pipelined_sweep_kernels_have_no_scratch(device_asm_variants, kernel, max_vgpr):
    res = _resources(device_asm_variants, kernel)
    assert len(res) == 4, sorted(res)   # <NT, OOP> x 2 x 2
    for name, r in res.items():
        assert r["private_seg_size"] == 0, (name, r)
        assert r["num_agpr"] == 0 and r["num_vgpr"] <= max_vgpr, (name, r)   # two waves per SIMD, nothing parked in AGPRs


def test_one_hop_decision_kernel_has_no_scratch(device_asm):
    """k_block_chain_t<true, 32> (round 3's decision kernel: in the product only for the opt-in one-hop exchange of the
    shards) keeps its register arrays in registers; the one-device instantiations live in the variants library."""
    res = _resources(device_asm, "k_block_chain_t")
    assert len(res) == 1 and all("Lb1ELi32E" in k for k in res), sorted(res)
    for name, r in res.items():
        assert r["private_seg_size"] == 0, (name, r)


def test_product_library_has_no_superseded_kernels(device_asm, device_asm_fused, device_asm_variants):
    """VERDICT r04, Next 7: the sweep kernels that lost their A/Bs are not compiled into liblpx.so; the variants library
    still has them (tests/test_gpu_variants.py checks their bits on the GPU)."""
    gone = ("k_sweep32_steady", "k_sweep32_dma", "k_sweep64_pull", "k_sweep64_pipe", "k_sweep64_mfmaI", "k_sweep64_mfma2_diag")
    for asm in (device_asm, device_asm_fused):
        for k in gone:
            assert not re.search(r"\.amdhsa_kernel \S*%s" % k, asm), k
    for k in ("k_sweep32_steady", "k_sweep32_dma", "k_sweep64_pull", "k_sweep64_pipe"):
        assert re.search(r"\.amdhsa_kernel \S*%s" % k, device_asm_variants), k


def _kernel_bodies(asm, kernel):
    """{mangled name: [instruction lines]} of every instantiation of `kernel` (labels and comments kept)."""
    out = {}
    for m in re.finditer(r"^(\S*%s\S*):\s*(?:;.*)?$" % kernel, asm, flags=re.M):
        end = asm.find("s_endpgm", m.end())
        out[m.group(1)] = [ln.strip() for ln in asm[m.end():end].splitlines()]
    return out


def _kernel_functions(asm, kernel):
    """like _kernel_bodies, but the WHOLE function (a kernel with early returns has several s_endpgm)."""
    out = {}
    for m in re.finditer(r"^(\S*%s\S*):\s*(?:;.*)?$" % kernel, asm, flags=re.M):
        end = asm.find(".Lfunc_end", m.end())
        out[m.group(1)] = [ln.strip() for ln in asm[m.end():end].splitlines()]
    return out


@pytest.mark.parametrize("kernel", ["k_sweep32_steady", "k_sweep64_pipe"])
def test_no_register_copy_sits_in_front_of_a_hand_written_wait(device_asm_variants, kernel):
    device_asm = device_asm_variants
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
def test_no_register_copy_reads_a_hand_issued_load_before_its_wait(device_asm_variants, kernel):
    device_asm = device_asm_variants
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
def test_lds_dma_sweep_kernels_have_no_scratch(device_asm_variants, kernel):
    device_asm = device_asm_variants
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
def test_pulled_tickets_are_not_touched_before_they_are_taken(device_asm_variants, kernel):
    device_asm = device_asm_variants
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
def test_decision_kernel_has_no_scratch_and_a_branch_free_ladder(device_asm, device_asm_fused):
    """k_block_chain2_t (the decision kernel of the blocked loop; one device and shards, 32- and 64-slot rings) in both
    compilations: one wave per SIMD, ONE window of live chunks in registers — no scratch in any instantiation (round 4's
    64-slot form held its whole ring: 404 registers, and spilled with the new ladder).  The pending-pivot ladder is inline
    asm: between an ASMSTART / ASMEND pair of a ladder chunk there are only the multiply-adds (fused: v_fma_f64; default:
    v_mul_f64 + v_add_f64), the eight v_cmp and the s_mov of EXEC — no branch, no select, no wait."""
    for asm, fused in ((device_asm, False), (device_asm_fused, True)):
        res = _resources(asm, "k_block_chain2_t")
        assert len(res) == 4, sorted(res)   # <32 | 64, 256, one device | shards>
        for name, r in res.items():
            assert r["private_seg_size"] == 0, (name, r)
        for name, lines in _kernel_functions(asm, "k_block_chain2_t").items():
            blocks, cur = [], None
            for ln in lines:
                if ln.startswith(";;#ASMSTART"):
                    cur = []
                elif ln.startswith(";;#ASMEND"):
                    if cur is not None:
                        blocks.append(cur)
                    cur = None
                elif cur is not None and ln:
                    cur.append(ln)
            arith = ("v_fma_f64",) if fused else ("v_mul_f64", "v_add_f64")
            ladders = [b for b in blocks if sum(1 for ln in b if ln.startswith(arith)) == (8 if fused else 16)]
            assert len(ladders) >= 32, (name, len(ladders))    # 8 chunks x (straight | from a start index) x 2 phases
            for b in ladders:
                assert all(ln.startswith(arith + ("v_cmp_ge_i32", "s_mov_b64")) for ln in b), (name, b)
            masked = [b for b in ladders if any(ln.startswith("v_cmp_ge_i32") for ln in b)]
            assert masked and all(sum(1 for ln in b if ln.startswith("v_cmp_ge_i32")) == 8 and b[0] == "s_mov_b64 %s, exec" % b[0].split()[1].rstrip(",")
                                  and b[-1].startswith("s_mov_b64 exec, ") for b in masked), name


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


@pytest.fixture(scope="module")
def device_asm_sweep128(tmp_path_factory):
    """scripts/micro/sweep_mfma128.hip (fused compilation + csrc/variants/): the only translation unit that instantiates
    k_sweep128_mfma until the engine dispatches to it."""
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm128") / "sweep_mfma128.s"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-ffp-contract=off", "-Wno-unused-result", "-DLPX_FUSED=1", "-DLPX_WITH_VARIANTS",
                           "-I", os.path.dirname(SRC), "-S", "--cuda-device-only",
                           os.path.join(ROOT, "scripts", "micro", "sweep_mfma128.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    return out.read_text()


def test_sweep_of_128_pivots_loop_as_designed(device_asm_sweep128):
    """k_sweep128_mfma (csrc/variants/, EXPERIMENTS 000.55): two workgroups per CU (<= 256 registers, <= 80 KiB of LDS, no
    scratch) and a tile loop as hand-counted: straight-line, two tiles of 128 MFMAs per round, per tile 16 + 16 loads, 16
    stores and ONE hand-issued ticket atomic, whose take waits behind exactly what was issued after it (vmcnt(49) = 32 loads +
    16 stores + the next pull); no other wait of the loop is stricter than a step's own operations (>= 48: the compiler's waits
    for the tile loaded a step earlier)."""
    asm = device_asm_sweep128
    res = _resources(asm, "k_sweep128_mfma")
    assert len(res) >= 1, "k_sweep128_mfma is not instantiated"
    for name, r in res.items():
        assert r["private_seg_size"] == 0 and r["num_vgpr"] + r["num_agpr"] <= 256, (name, r)
    lds = re.findall(r"\.amdhsa_group_segment_fixed_size (\d+)", "".join(
        asm[m.start():m.start() + 4000] for m in re.finditer(r"\.amdhsa_kernel \S*k_sweep128_mfma", asm)))
    assert lds and all(int(x) <= 80 * 1024 for x in lds), lds
    for name, lines in _kernel_functions(asm, "k_sweep128_mfma").items():
        lines = [ln for ln in lines if ln and not ln.startswith(";")]
        back = [k for k, ln in enumerate(lines) if re.match(r"s_cbranch_scc[01] \.LBB\d+_\d+", ln)]
        labels = {m.group(1): k for k, ln in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
        loops = [(labels[ln.split()[-1]], k) for k in back for ln in [lines[k]] if labels.get(ln.split()[-1], k) < k]
        assert loops, name
        head, tail = max(loops, key=lambda ht: sum(1 for ln in lines[ht[0]:ht[1]] if ln.startswith("v_mfma_f64_16x16x4")))
        body = lines[head:tail]
        assert sum(1 for ln in body if ln.startswith("v_mfma_f64_16x16x4")) == 256, name
        assert sum(1 for ln in body if ln.startswith("global_atomic_add")) == 2, name
        assert not any(ln.startswith(("s_cbranch", "s_branch", "s_barrier")) for ln in body), name
        assert sum(1 for ln in body if ln.startswith("buffer_load_dwordx2")) == 32, name      # two tiles of C
        assert sum(1 for ln in body if ln.startswith("buffer_load_dwordx4")) == 32, name      # two tiles of A operands
        assert sum(1 for ln in body if ln.startswith("buffer_store_dwordx2")) == 32, name
        assert not any(ln.startswith(("global_load", "global_store", "flat_", "scratch_", "v_lshl_add_u64")) for ln in body), name
        # the order of the memory operations: pull, take (vmcnt(49)), 32 loads, the tile's arithmetic, 16 stores — twice
        seq = []
        for ln in body:
            if ln.startswith("global_atomic_add"):
                seq.append("A")
            elif ln.startswith("buffer_load"):
                seq.append("L")
            elif ln.startswith("buffer_store"):
                seq.append("S")
        assert "".join(seq) == ("A" + "L" * 32 + "S" * 16) * 2, (name, "".join(seq))
        waits = [int(x) for ln in body for x in re.findall(r"vmcnt\((\d+)\)", ln)]
        assert waits.count(49) >= 2 and min(waits) >= 48, (name, sorted(set(waits)))
        first_wait_after_atomic = [next(int(re.findall(r"vmcnt\((\d+)\)", ln2)[0]) for ln2 in body[k:] if "vmcnt(" in ln2)
                                   for k, ln in enumerate(body) if ln.startswith("global_atomic_add")]
        assert first_wait_after_atomic == [49, 49], (name, first_wait_after_atomic)
