"""ctypes binding of ``libpdlp_hip.so`` (C ABI: ``include/pdlp_hip.h``).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception is
raised.  Build it with ``torchpdlp_amd/csrc/build.sh`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDLP_LIB") or os.path.join(_HERE, "libpdlp_hip.so")   # PDLP_LIB: profiling/ablation builds (empty = unset)

ABI_VERSION = 17
PDLP_F32, PDLP_F64, PDLP_MIXED = 0, 1, 2
CUR, AVG, PREV = 0, 1, 2
(BUF_X_CUR, BUF_X_PREV, BUF_XBAR, BUF_X_AVG, BUF_Y_CUR, BUF_Y_PREV, BUF_Y_AVG, BUF_RED, BUF_X_SUM, BUF_Y_SUM,
 BUF_SCALARS, BUF_DX, BUF_DY, BUF_LAM_PREV, BUF_GDX, BUF_GDY) = range(16)
NRED, NSCAL = 8, 16
OPT_RUNNING_KKT, OPT_KTY_REUSE, OPT_GRAPH, OPT_SPLIT_SLOTS, OPT_PRODUCER_PIECES, OPT_BEGIN_INLINE, OPT_PEER_EXCHANGE, OPT_PEER_TIMEOUT_MS, OPT_PEER_LOCAL_FIRST, OPT_PEER_PUSH = range(10)
PEER_INFO_BYTES, PEER_LOOPBACK, PEER_LOOPBACK_HOST = 256, 1, 2
# indices into the scalar block
S_ETA, S_OMEGA, S_THETA, S_TAU, S_SIGMA, S_WPEND, S_ETASUM, S_K, S_INV1PT, S_ACCEPT, S_ETABAR, S_DEN, S_ETASUM_PREV = range(13)


class PdlpProblem(C.Structure):
    """mirror of ``struct pdlp_problem``"""
    _fields_ = [("dtype", C.c_int32), ("device", C.c_int32), ("m", C.c_int64), ("n", C.c_int64), ("m_ineq", C.c_int64),
                ("row0", C.c_int64), ("row1", C.c_int64), ("col0", C.c_int64), ("col1", C.c_int64),
                ("K_rowptr", C.c_void_p), ("K_colidx", C.c_void_p), ("K_val", C.c_void_p),
                ("KT_rowptr", C.c_void_p), ("KT_colidx", C.c_void_p), ("KT_val", C.c_void_p),
                ("c", C.c_void_p), ("l", C.c_void_p), ("u", C.c_void_p), ("q", C.c_void_p),
                ("d_col", C.c_void_p), ("d_row", C.c_void_p), ("stream", C.c_void_p)]


class PdlpTiles(C.Structure):
    """mirror of ``struct pdlp_tiles``"""
    _fields_ = [("lw", C.c_int32), ("rpt", C.c_int32), ("cap", C.c_int32), ("nblk", C.c_int32), ("npanel", C.c_int32),
                ("groups", C.c_int32), ("idx", C.c_void_p), ("val", C.c_void_p), ("tile_ptr", C.c_void_p), ("blk_base", C.c_void_p), ("cnt", C.c_void_p),
                ("rem_rows_n", C.c_int32), ("rem_segs_n", C.c_int32), ("rem_rows", C.c_void_p), ("rem_rptr", C.c_void_p),
                ("rem_sptr", C.c_void_p), ("rem_col", C.c_void_p), ("rem_val", C.c_void_p), ("rem_work", C.c_void_p),
                ("rem_extra", C.c_void_p), ("rem_extra_f32", C.c_void_p)]


# every symbol include/pdlp_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
_I, _I64, _D, _P = C.c_int, C.c_int64, C.c_double, C.c_void_p
SIGNATURES = {
    "pdlp_strerror": (C.c_char_p, [_I]),
    "pdlp_abi_version": (_I, []),
    "pdlp_workspace_bytes": (_I, [C.POINTER(PdlpProblem), C.POINTER(_I64)]),
    "pdlp_create": (_I, [C.POINTER(_H), C.POINTER(PdlpProblem), _P, _I64]),
    "pdlp_destroy": (None, [_H]),
    "pdlp_attach_tiles": (_I, [_H, _I, C.POINTER(PdlpTiles)]),
    "pdlp_schedule_info": (_I, [_H, _I, C.POINTER(C.c_int32), C.POINTER(_P)]),
    "pdlp_attach_sorted": (_I, [_H, _I, _P, _P, _P]),
    "pdlp_buffer_ptr": (_I, [_H, _I, C.POINTER(_P)]),
    "pdlp_set_iterate": (_I, [_H, _P, _P]),
    "pdlp_get_iterate": (_I, [_H, _I, _P, _P]),
    "pdlp_set_step": (_I, [_H, _D, _D, _D, _I64]),
    "pdlp_set_omega": (_I, [_H, _D]),
    "pdlp_get_scalars": (_I, [_H, C.POINTER(_D)]),
    "pdlp_primal_half": (_I, [_H, _I]),
    "pdlp_dual_half": (_I, [_H, _I]),
    "pdlp_primal_half_piece": (_I, [_H, _I, _I, _I]),
    "pdlp_dual_half_piece": (_I, [_H, _I, _I, _I]),
    "pdlp_primal_half_begin": (_I, [_H]),
    "pdlp_dual_half_begin": (_I, [_H, _I]),
    "pdlp_split_info": (_I, [_H, _I, C.POINTER(C.c_int32)]),
    "pdlp_tile_limits": (_I, [_H, C.POINTER(C.c_int32)]),
    "pdlp_adaptive_retry": (_I, [_H]),
    "pdlp_adaptive_reduce": (_I, [_H]),
    "pdlp_adaptive_update": (_I, [_H]),
    "pdlp_iterate": (_I, [_H, _I, _I]),
    "pdlp_set_exchange_chunks": (_I, [_H, _I]),
    "pdlp_set_option": (_I, [_H, _I, _I64]),
    "pdlp_exchange_plan": (_I, [_H, _I, C.POINTER(C.c_int32), C.POINTER(_I64)]),
    "pdlp_half_chunk": (_I, [_H, _I, _I]),
    "pdlp_comm_load": (_I, [C.c_char_p]),
    "pdlp_comm_unique_id": (_I, [C.c_char_p, _P]),
    "pdlp_comm_init": (_I, [_H, C.c_char_p, _P, _I, _I]),
    "pdlp_peer_export": (_I, [_H, _P]),
    "pdlp_peer_connect": (_I, [_H, _I, _I, _P, _I]),
    "pdlp_peer_status": (_I, [_H, C.POINTER(C.c_int32)]),
    "pdlp_peer_close": (_I, [_H]),
    "pdlp_comm_all_gather": (_I, [_H, _I]),
    "pdlp_comm_all_reduce_red": (_I, [_H]),
    "pdlp_set_delta": (_I, [_H, _I]),
    "pdlp_refresh_products": (_I, [_H]),
    "pdlp_set_anchors": (_I, [_H, _P, _P]),
    "pdlp_delta_state": (_I, [_H, C.POINTER(C.c_int32)]),
    "pdlp_fixed_advance": (_I, [_H, _I]),
    "pdlp_flush_average": (_I, [_H, _I]),
    "pdlp_compute_average": (_I, [_H]),
    "pdlp_kkt_local": (_I, [_H, _I, _I]),
    "pdlp_kkt_finish": (_I, [_H, _D, C.POINTER(_D)]),
    "pdlp_restart": (_I, [_H, _I]),
    "pdlp_restart_distance_local": (_I, [_H]),
    "pdlp_mark_restart_point": (_I, [_H]),
    "pdlp_read_red": (_I, [_H, C.POINTER(_D)]),
    "pdlp_infeas_reset": (_I, [_H]),
    "pdlp_infeas_begin": (_I, [_H]),
    "pdlp_infeas_local": (_I, [_H, _D]),
    "pdlp_infeas_finish": (_I, [_H, _D, C.POINTER(C.c_int32), C.POINTER(_D)]),
    "pdlp_mv_steps": (_I, [_H, _I, _I, _D, _D, _D, _P, _P, _P]),
    "pdlp_mv_gap": (_I, [_H, _I, _P, _P, _P, C.POINTER(_D)]),
    "pdlp_mv_product": (_I, [_H, _I, _P, _P]),
    "pdlp_mv_combine": (_I, [_I, _I64, _I, _P, _P, _I, _P, _P]),
    "pdlp_spmv": (_I, [_H, _I, _P, _P]),
    "pdlp_power_iteration": (_I, [_H, _P, _I, _P, _P, C.POINTER(_D)]),
    "pdlp_probe_stream_read": (_I, [_P, _I64, _I, _P, C.POINTER(_D)]),
    "pdlp_trace_enable": (_I, [_I]),
    "pdlp_range_push": (_I, [C.c_char_p, _P]),
    "pdlp_range_pop": (_I, [_P]),
    "pdlp_probe_gather": (_I, [_P, _I64, _I64, _I, _P, C.POINTER(_D)]),
    "pdlp_csr_row_scale_factors": (_I, [_I, _I64, _P, _P, _D, _P, _P]),
    "pdlp_csr_div_rows": (_I, [_I, _I64, _P, _P, _P, _P]),
    "pdlp_csr_div_cols": (_I, [_I, _I64, _P, _P, _P, _P]),
    "pdlp_vec_muldiv": (_I, [_I, _I64, _P, _P, _I, _P]),
    "pdlp_vec_project_lambda": (_I, [_I, _I64, _P, _P, _P, _P, _P]),
    "pdlp_vec_max_dev_from_one": (_I, [_I, _I64, _P, _P, C.POINTER(_D), _P]),
    "pdlp_vec_sqdist": (_I, [_I, _I64, _P, _P, _P, C.POINTER(_D), _P]),
}

_lib = None


class PdlpError(RuntimeError):
    pass


def load():
    """Load the HIP library (no GPU needed to load it; calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PdlpError(f"{LIB_PATH} not found: the HIP extension is not built "
                        f"(run torchpdlp_amd/csrc/build.sh). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pdlp_abi_version() != ABI_VERSION:
        raise PdlpError(f"libpdlp_hip.so ABI {lib.pdlp_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


class trace_range:
    """``with trace_range("restart check", stream):`` -- a roctx range (rocprofv3 --marker-trace) when tracing is on
    (``PDLP_ROCTX=1|2`` in the environment, or ``pdlp_trace_enable``); nothing otherwise.  ``stream``: a torch stream or None."""
    level = None

    def __init__(self, name: str, stream=None):
        self.name, self.stream = name.encode(), (None if stream is None else stream.cuda_stream)

    @classmethod
    def enabled(cls) -> bool:
        if cls.level is None:
            lvl = int(os.environ.get("PDLP_ROCTX", "0") or 0)
            cls.level = lvl if lvl > 0 and os.path.exists(LIB_PATH) and load().pdlp_trace_enable(lvl) == 0 else 0
        return cls.level > 0

    def __enter__(self):
        self.on = self.enabled()
        if self.on:
            load().pdlp_range_push(self.name, self.stream)
        return self

    def __exit__(self, *exc):
        if self.on:
            load().pdlp_range_pop(self.stream)
        return False


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pdlp_strerror(rc).decode()
        raise PdlpError(f"{what or 'libpdlp_hip'} failed: {msg} (code {rc})")
