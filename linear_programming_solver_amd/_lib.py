"""ctypes binding of liblpx.so (include/lpx.h).  The product path has NO CPU fallback: if the HIP library
is missing or cannot be loaded this module raises, loudly, at first use."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LPX_LIB_PATH: kernel-variant experiments of scripts/ load another build of the same library (still no fallback)
LIB_PATH = os.environ.get("LPX_LIB_PATH") or os.path.join(_HERE, "liblpx.so")

# lpx_status (include/lpx.h)
OPTIMAL, UNBOUNDED, INFEASIBLE, AUX_UNBOUNDED, NO_DEGENERATE_PIVOT, BAD_ARGUMENT, RESTORE_INDEX_FAULT, \
    DEVICE_ERROR, DIVIDE_BY_ZERO, PIVOT_LIMIT = range(10)
CAND_HEADER = 8
PRICING = {"reference": 0, "first-positive": 0, "dantzig": 1, 0: 0, 1: 1}

# lpx_option (include/lpx.h)
OPTIONS = {"block": 0, "chain": 1, "overlap": 2, "overlap_serial": 3, "overlap_mask": 4, "chain_wgs": 5,
           "chain_fences": 6, "sweep_rows": 7, "nt": 8, "batch": 9, "chain_trace": 10, "update_u": 11,
           "update_rows": 12, "a2_offset": 13, "sweep_form": 14, "multi_onehop": 15, "sweep_cus": 16, "chain_cus": 17,
           "fused": 18, "chain_form": 19, "fixup_side": 20}

# Arithmetic of the handles the host classes create when the caller does not say (option "fused" / LPSolver(fused=...)):
# None = the library's choice (LPX_OPT_FUSED = 2: by size — fused multiply-add updates on an unsharded tableau of 0.5 GiB
# and more, otherwise product and difference of every update rounded separately, as the reference rounds them),
# False = always the two roundings, True = always fused.  See set_default_arithmetic().
DEFAULT_FUSED = None


def set_default_arithmetic(mode):
    """"auto" (the library's choice by size; default), "plain" or "fused": what LPState / LPMulti / LPSolver /
    HipShardEngine select for new handles unless told otherwise.  Returns the previous mode."""
    global DEFAULT_FUSED
    if mode not in ("auto", "plain", "fused"):
        raise ValueError('arithmetic mode is "auto", "plain" or "fused"')
    prev = "auto" if DEFAULT_FUSED is None else "fused" if DEFAULT_FUSED else "plain"
    DEFAULT_FUSED = None if mode == "auto" else mode == "fused"
    return prev


dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class SolveResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("phase1_used", C.c_int32),
        ("objective", C.c_double),
        ("objective_rounded", C.c_double),
        ("objective_text", C.c_char * 64),
        ("pivots_phase1", C.c_int64),
        ("pivots_phase2", C.c_int64),
        ("x0_slot", C.c_int32),
        ("reserved", C.c_int32),
        ("seconds_total", C.c_double),
        ("seconds_pivots", C.c_double),
    ]


class StateInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "block", "chain_wgs", "chain_wgs_requested", "chain_resident_max", "chain_blocks_per_cu",
        "chain_stream_masked", "chain_xcd_mask", "sweep_xcd_mask", "overlapped", "nontemporal", "sweep_rows",
        "sweep_kernel", "multi_onehop", "sweep_clock_mhz", "sweep_cus", "arith_fused")]


class SolveOptions(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("has_variable_names", C.c_int32),
        ("max_pivots", C.c_int64),
        ("restore_order", ip),
        ("perm_out", ip),
        ("x_out", dp),
        ("keep_state", C.POINTER(C.c_void_p)),
        ("pricing", C.c_int32),
        ("restore_order_len", C.c_int32),
        ("fused", C.c_int32),
        ("reserved", C.c_int32),
    ]


# every symbol include/lpx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("lpx_status_message", C.c_char_p, [C.c_int]),
    ("lpx_last_error", C.c_char_p, []),
    ("lpx_abi_version", C.c_int, []),
    ("lpx_device_count", C.c_int, []),
    ("lpx_state_create", C.c_int, [C.c_int32, C.c_int32, dp, C.c_int64, dp, dp, C.c_double, ip, C.c_int32,
                                   C.c_int32, C.c_int, C.POINTER(C.c_void_p)]),
    ("lpx_state_create_from_device", C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_double, ip, C.c_int32, C.c_int32, C.c_int,
                                               C.POINTER(C.c_void_p)]),
    ("lpx_state_destroy", None, [C.c_void_p]),
    ("lpx_state_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("lpx_state_use_masked_stream", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    ("lpx_state_set_pricing", C.c_int, [C.c_void_p, C.c_int32]),
    ("lpx_state_set_block", C.c_int, [C.c_void_p, C.c_int32]),
    ("lpx_state_get_block", C.c_int, [C.c_void_p]),
    ("lpx_state_set_option", C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    ("lpx_state_get_option", C.c_int, [C.c_void_p, C.c_int32, i64p]),
    ("lpx_state_get_info", C.c_int, [C.c_void_p, C.POINTER(StateInfo)]),
    ("lpx_sweep_kernel_name", C.c_char_p, [C.c_int32]),
    ("lpx_state_read_chain_trace", C.c_int, [C.c_void_p, i64p, C.c_int32, ip]),
    ("lpx_state_read_chain_trace_fine", C.c_int, [C.c_void_p, i64p, C.c_int32, ip, ip]),
    ("lpx_get_entering", C.c_int, [C.c_void_p, ip]),
    ("lpx_get_leaving", C.c_int, [C.c_void_p, C.c_int32, ip, dp]),
    ("lpx_pivot", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("lpx_simplex_loop", C.c_int, [C.c_void_p, C.c_int64, i64p, ip, ip]),
    ("lpx_state_read", C.c_int, [C.c_void_p, dp, C.c_int64, dp, dp, dp, ip]),
    ("lpx_state_dims", C.c_int, [C.c_void_p, ip, ip, ip, ip]),
    ("lpx_state_checksum", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("lpx_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("lpx_profile_read", C.c_int, [C.c_void_p, i64p, dp]),
    ("lpx_shard_begin", C.c_int, [C.c_void_p, C.c_int64, C.c_int32]),
    ("lpx_shard_propose", C.c_int, [C.c_void_p, C.c_void_p]),
    ("lpx_shard_commit", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ("lpx_shard_probe", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ("lpx_shard_set_comm_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("lpx_shard_set_pipeline", C.c_int, [C.c_void_p, C.c_int32]),
    ("lpx_shard_peek", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("lpx_shard_decide", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("lpx_shard_update", C.c_int, [C.c_void_p, C.c_int32]),
    ("lpx_shard_block_peek", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ("lpx_shard_block_decide", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("lpx_shard_block_sweep", C.c_int, [C.c_void_p, C.c_int32]),
    ("lpx_shard_poll", C.c_int, [C.c_void_p, i64p, ip]),
    ("lpx_multi_create", C.c_int, [C.c_int32, C.c_int32, dp, C.c_int64, dp, dp, C.c_double, ip, ip, C.c_int32,
                                   C.POINTER(C.c_void_p)]),
    ("lpx_multi_destroy", None, [C.c_void_p]),
    ("lpx_multi_set_option", C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    ("lpx_multi_set_pricing", C.c_int, [C.c_void_p, C.c_int32]),
    ("lpx_multi_get_entering", C.c_int, [C.c_void_p, ip]),
    ("lpx_multi_get_leaving", C.c_int, [C.c_void_p, C.c_int32, ip, dp]),
    ("lpx_multi_pivot", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("lpx_multi_simplex_loop", C.c_int, [C.c_void_p, C.c_int64, i64p, ip, ip]),
    ("lpx_multi_read", C.c_int, [C.c_void_p, dp, C.c_int64, dp, dp, dp, ip]),
    ("lpx_multi_checksum", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("lpx_multi_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("lpx_multi_profile_read", C.c_int, [C.c_void_p, C.c_int32, i64p, dp]),
    ("lpx_multi_get_info", C.c_int, [C.c_void_p, C.POINTER(StateInfo)]),
    ("lpx_solve_multi", C.c_int, [C.c_int32, C.c_int32, dp, C.c_int64, dp, dp, C.c_int32, C.POINTER(SolveOptions),
                                  ip, C.c_int32, C.POINTER(SolveResult)]),
    ("lpx_solve", C.c_int, [C.c_int32, C.c_int32, dp, C.c_int64, dp, dp, C.c_int32, C.POINTER(SolveOptions),
                            C.POINTER(SolveResult)]),
    ("lpx_restore_initial_lp", C.c_int, [C.c_void_p, dp, C.c_int32, C.c_int32, ip, C.c_int32]),
    ("lpx_java_default_name_order", C.c_int, [C.c_int32, ip]),
    ("lpx_transpose", C.c_int, [C.c_int32, C.c_int32, dp, C.c_int64, dp, C.c_int64, C.c_int]),
]

_lib = None


def lib():
    """Load liblpx.so and bind every declared symbol; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "liblpx.so (the HIP extension) is missing at %s: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C linear_programming_solver_amd/csrc`. "
            "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


def last_error():
    return lib().lpx_last_error().decode()


def status_message(status):
    return lib().lpx_status_message(int(status)).decode()
