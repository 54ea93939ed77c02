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


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "lpx_kernels.s"
    # the flags of csrc/Makefile that matter for code generation
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-ffp-contract=off", "-Wno-unused-result", "-S", "--cuda-device-only", SRC, "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


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
