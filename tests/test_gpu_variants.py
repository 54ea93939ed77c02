"""The superseded sweep kernels of csrc/variants/ (not part of liblpx.so: `make variants` builds
gpurun_variants/liblpx_variants.so, which scripts/ and the micro benchmarks load for same-box A/Bs): every one of them must
still leave the bits of the oracle — an A/B against a kernel that computes something else says nothing.  The module swaps
the host package's library for the variants library and back; the tests are the ones these kernels had in
tests/test_gpu_parity.py while they were part of the product."""
import os

import pytest

from tests.test_gpu_parity import _timed_form_vs_oracle, assert_state_bits_equal, dense_lp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = os.path.join(ROOT, "gpurun_variants", "liblpx_variants.so")


@pytest.fixture(scope="module")
def variants_library():
    """linear_programming_solver_amd bound to the variants library for this module (handles keep the library they were
    created with; the product library is bound again afterwards)."""
    if not os.path.exists(VARIANTS):
        pytest.skip("gpurun_variants/liblpx_variants.so is not built (make -C linear_programming_solver_amd/csrc variants)")
    import ctypes as C
    from linear_programming_solver_amd import _lib
    _lib.lib()
    keep = (_lib._lib, _lib.LIB_PATH)
    L = C.CDLL(VARIANTS)
    for name, restype, argtypes in _lib.SYMBOLS:
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib._lib, _lib.LIB_PATH = L, VARIANTS
    yield L
    _lib._lib, _lib.LIB_PATH = keep


@pytest.fixture(scope="module")
def lps(arith, variants_library):
    from tests.conftest import package_in_mode
    pkg = package_in_mode(arith)
    yield pkg
    pkg.set_default_arithmetic("auto")


@pytest.fixture(scope="module")
def oracle(arith):
    from oracle import pyoracle
    from tests.conftest import ArithOracle
    pyoracle.build()
    pyoracle.lib()
    return ArithOracle(pyoracle, arith)


@pytest.mark.parametrize("form,name", [(1, "k_sweep32_steady"), (2, "k_sweep32_dma")])
def test_cfg3_superseded_sweeps_of_blocks_of_32_vs_fp64_oracle(lps, oracle, form, name):
    """Round 2's register-staged sweep and round 3's LDS-DMA sweep with runs of rows, each through two full blocks and a tail
    at cfg3 against the oracle of the arithmetic mode."""
    info = _timed_form_vs_oracle(lps, oracle, 8192, 16384, (75, 60), options={"sweep_form": form, "block": 32})
    assert info["block"] == 32 and info["sweep_kernel_name"] == name


@pytest.mark.parametrize("shape", [(1000, 2100), (4100, 1024), (2052, 4100)])
@pytest.mark.parametrize("block", [40, 64])
def test_pair_of_waves_sweep_on_ragged_shapes(lps, oracle, shape, block):
    """k_sweep64_pull (round 3: blocks of 33..64, a pair of waves per 128-column sub-strip) on shapes with a partial last
    strip, fewer batches than workers and partly filled blocks."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=11 * m + n)
    st = lps.LPState(A, b, c, block=block, options={"sweep_form": 2})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (block - 1, 2 * block + 5, block - 1):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=8)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, block, budget)
        assert_state_bits_equal(st.read(), ref.read(), "block %d budget %d of %s" % (block, budget, shape))
    if status == 9:
        assert st.info()["sweep_kernel_name"] == "k_sweep64_pull", st.info()
    st.close()


@pytest.mark.parametrize("shape", [(1024, 2112), (2048, 4100)])
@pytest.mark.parametrize("block", [40, 64])
def test_first_mfma_sweep_with_16_row_tiles(lps, oracle, arith, shape, block):
    """k_sweep64_mfma (round 4, first version: one wave per SIMD, B operands in registers; fused mode, sweep_form = 4)."""
    if arith != "fused":
        pytest.skip("the matrix-core sweeps exist in the fused compilation only")
    m, n = shape
    A, b, c = dense_lp(m, n, seed=17 * m + n)
    st = lps.LPState(A, b, c, block=block, options={"sweep_form": 4})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (block - 1, 2 * block + 5, block - 1):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=8)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, block, budget)
        assert_state_bits_equal(st.read(), ref.read(), "block %d budget %d of %s" % (block, budget, shape))
    if status == 9:
        assert st.info()["sweep_kernel_name"] == "k_sweep64_mfma", st.info()
    st.close()


def test_round3_decision_kernel_on_one_device(lps, oracle):
    """k_block_chain_t<false, KB> (chain_form = 0: round 3's decision kernel, instantiated for one device by the variants library
    only) through the overlapped loop: three blocks and a tail."""
    m, n = 2048, 4096
    A, b, c = dense_lp(m, n, seed=99)
    st = lps.LPState(A, b, c, block=32, options={"chain_form": 0})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (100, 45):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=8)
        assert (status, pivots) == (want["status"], want["pivots"]), budget
        assert_state_bits_equal(st.read(), ref.read(), "chain_form 0, budget %d" % budget)
    st.close()


def test_sweep_of_128_pivots_matches_two_passes_of_64_bit_for_bit():
    """k_sweep128_mfma (csrc/variants/: a block of up to 128 pivots in one pass on the matrix cores, EXPERIMENTS 000.55) is not
    reachable through the C ABI yet; `make variants` builds scripts/micro/sweep_mfma128.hip, which runs it on a synthetic ring
    with 128 / 100 / 64 / 40 valid pivots and compares every entry with two passes of the product's k_sweep64_mfma2 (whose bits
    the tests of tests/test_gpu_parity.py pin to the oracle's 64 sequential fused updates, LPState.java:162).  Exit code 0 = no
    entry differs."""
    import subprocess
    exe = os.path.join(ROOT, "gpurun_variants", "sweep_mfma128")
    if not os.path.exists(exe):
        pytest.skip("gpurun_variants/sweep_mfma128 is not built (make -C linear_programming_solver_amd/csrc variants)")
    out = subprocess.run([exe, "1024", "1024", "2", "32", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("np ")]
    assert len(lines) == 4 and all(ln.rstrip().endswith("entries that differ: 0") for ln in lines), out.stdout
