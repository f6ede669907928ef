"""ctypes loader for libfwx.so (the C ABI of include/fwx.h).  Fails loudly: there is no fallback."""
import ctypes
import os

# Multi-process / multi-device GPU work on this pool needs dmabuf IPC: without it RCCL (and any
# device-memory sharing across processes) fails in hipIpcGetMemHandle ("invalid argument").  The HIP
# runtime reads the variable when it initialises, so it is set HERE, at import, before the first HIP
# call any binding of this package can make -- a host that binds libfwx directly (C, Haskell) exports
# it itself (include/fwx.h FWX_XCHG_RCCL).  setdefault: an explicit setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PKG = os.path.dirname(os.path.abspath(__file__))
# FWX_LIB_PATH: developer override, e.g. an experimental build of the same ABI
LIB_PATH = os.environ.get("FWX_LIB_PATH") or os.path.join(PKG, "libfwx.so")

FWX_OK = 0
FWX_ERR_INVALID = -1
FWX_ERR_NO_DEVICE = -2
FWX_ERR_HIP = -3
FWX_ERR_OOM = -4
FWX_ERR_CYCLE = -5
FWX_ERR_CAPACITY = -6
FWX_ERR_UNSUPPORTED = -7
FWX_ERR_RCCL = -8
FWX_ERR_INTERNAL = -9

FWX_ABI_VERSION = 3
FWX_F32, FWX_F64 = 0, 1
FWX_ENGINE_AUTO, FWX_ENGINE_PERK, FWX_ENGINE_FUSED = 0, 1, 2
FWX_UPDATE_SHARDS = 256
FWX_FUSED_BLOCK = 64
FWX_FLAG_NONNEG = 1
FWX_XCHG_AUTO, FWX_XCHG_PEER, FWX_XCHG_RCCL = 0, 1, 2
FWX_XCHG_CALLBACK = 3
FWX_MAX_PARTS = 32

c_i32 = ctypes.c_int32
c_vp = ctypes.c_void_p
# int exchange(void *ctx, int32 k0, int32 bt, int32 owner, void *w, int32 *wh, int64 count, void *stream)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, ctypes.c_int64, c_vp)


class FwxOpts(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", c_i32), ("engine", c_i32),
                ("k_begin", c_i32), ("k_end", c_i32), ("block", c_i32), ("serpentine", c_i32),
                ("updates_out", ctypes.POINTER(ctypes.c_uint64)), ("stream", c_vp),
                ("use_stream", c_i32), ("reserved0", c_i32)]


class FwxSlab(ctypes.Structure):
    _fields_ = [("n", c_i32), ("row0", c_i32), ("rows", c_i32), ("dtype", c_i32),
                ("rate", c_vp), ("next", c_vp), ("hops", c_vp)]


class FwxPivots(ctypes.Structure):
    _fields_ = [("k_begin", c_i32), ("k_end", c_i32), ("rate", c_vp), ("hops", c_vp),
                ("stride", ctypes.c_int64), ("next", c_vp)]


class FwxTrace(ctypes.Structure):
    _fields_ = [("last", c_vp), ("at_col", c_vp), ("at_row", c_vp)]


class FwxFusedScratch(ctypes.Structure):
    _fields_ = [("col_rate", c_vp), ("col_next", c_vp), ("col_hops", c_vp)]


class FwxMultiTiming(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("steps", c_i32), ("pivots_per_step", c_i32),
                ("partitions", c_i32), ("bulk_us", ctypes.c_float), ("bulk_mean_us", ctypes.c_float),
                ("lookahead_us", ctypes.c_float), ("panel_us", ctypes.c_float),
                ("exchange_us", ctypes.c_float), ("chain_us", ctypes.c_float),
                ("chain_over_bulk", ctypes.c_float)]


class FwxError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        msg = lib().fwx_strerror(status).decode()
        extra = ""
        if status == FWX_ERR_HIP:
            extra = " (hipError_t %d)" % lib().fwx_last_hip_error()
        super().__init__("%s: %s%s" % (what, msg, extra))


# Every symbol include/fwx.h declares, with its ctypes signature.
SIGNATURES = {
    "fwx_abi_version": (ctypes.c_int, []),
    "fwx_device_count": (ctypes.c_int, []),
    "fwx_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "fwx_last_hip_error": (ctypes.c_int, []),
    "fwx_hip_versions": (ctypes.c_int, [ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "fwx_test_fail_after": (ctypes.c_int, [c_i32]),
    "fwx_solve_f64": (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, ctypes.POINTER(FwxOpts)]),
    "fwx_solve_f32": (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, ctypes.POINTER(FwxOpts)]),
    "fwx_follow_path": (ctypes.c_int, [c_i32, c_vp, c_i32, c_i32, c_vp, c_i32]),
    "fwx_matrix_create": (ctypes.c_int, [ctypes.POINTER(c_vp), c_i32, c_i32, c_i32, c_i32, c_i32]),
    "fwx_matrix_upload": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "fwx_matrix_solve": (ctypes.c_int, [c_vp, ctypes.POINTER(FwxOpts)]),
    "fwx_matrix_download": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "fwx_matrix_query": (ctypes.c_int, [c_vp, c_i32, c_i32, ctypes.POINTER(ctypes.c_double), c_vp,
                                        c_i32]),
    "fwx_matrix_destroy": (ctypes.c_int, [c_vp]),
    "fwx_matrix_keep_input": (ctypes.c_int, [c_vp]),
    "fwx_matrix_patch_input": (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "fwx_matrix_enable_resume": (ctypes.c_int, [c_vp, c_i32]),
    "fwx_matrix_resume_bytes": (ctypes.c_int, [c_vp, c_i32, ctypes.POINTER(ctypes.c_uint64)]),
    "fwx_device_memory": (ctypes.c_int, [c_i32, ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.POINTER(ctypes.c_uint64)]),
    "fwx_matrix_resolve": (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(FwxOpts),
                                          ctypes.POINTER(c_i32)]),
    "fwx_matrix_enable_path_log": (ctypes.c_int, [c_vp]),
    "fwx_matrix_path_log_count": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "fwx_matrix_query_exact": (ctypes.c_int, [c_vp, c_i32, c_i32, ctypes.POINTER(ctypes.c_double),
                                              c_vp, c_i32]),
    "fwx_matrix_query_exact_batch": (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32]),
    "fwx_matrix_create_multi": (ctypes.c_int, [ctypes.POINTER(c_vp), c_i32, c_i32, c_i32, c_i32, c_i32,
                                               ctypes.POINTER(c_i32), c_i32]),
    "fwx_matrix_create_part": (ctypes.c_int, [ctypes.POINTER(c_vp), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                              c_vp, c_vp]),
    "fwx_matrix_part_rows": (ctypes.c_int, [c_vp, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "fwx_matrix_domain_bits": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32)]),
    "fwx_matrix_set_domain": (ctypes.c_int, [c_vp, c_i32]),
    "fwx_matrix_parts": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32)]),
    "fwx_matrix_set_timing": (ctypes.c_int, [c_vp, c_i32]),
    "fwx_matrix_get_timing": (ctypes.c_int, [c_vp, ctypes.POINTER(FwxMultiTiming)]),
    "fwx_matrix_comm_ranks": (ctypes.c_int, [c_vp]),
    "fwx_solve_multi_f64": (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, c_i32, ctypes.POINTER(c_i32), c_i32,
                                           ctypes.POINTER(FwxOpts)]),
    "fwx_solve_multi_f32": (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, c_i32, ctypes.POINTER(c_i32), c_i32,
                                           ctypes.POINTER(FwxOpts)]),
    "fwx_dev_relax": (ctypes.c_int, [ctypes.POINTER(FwxSlab), ctypes.POINTER(FwxPivots), c_i32,
                                     c_vp, c_vp]),
    "fwx_dev_relax_skip": (ctypes.c_int, [ctypes.POINTER(FwxSlab), ctypes.POINTER(FwxPivots), c_i32,
                                          c_vp, c_i32, c_i32, c_vp]),
    "fwx_dev_panel": (ctypes.c_int, [ctypes.POINTER(FwxSlab), c_vp, c_vp, c_vp, c_vp]),
    "fwx_dev_solve": (ctypes.c_int, [ctypes.POINTER(FwxSlab), ctypes.POINTER(FwxOpts)]),
    "fwx_dev_follow_paths": (ctypes.c_int, [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                            c_vp, c_i32, c_vp]),
    "fwx_dev_panel_snap": (ctypes.c_int, [ctypes.POINTER(FwxSlab), c_vp, c_vp, ctypes.POINTER(FwxTrace),
                                          c_vp]),
    "fwx_dev_relax_fused": (ctypes.c_int, [ctypes.POINTER(FwxSlab), ctypes.POINTER(FwxPivots),
                                           ctypes.POINTER(FwxFusedScratch), ctypes.POINTER(FwxTrace),
                                           c_vp, c_i32, c_vp]),
    "fwx_dev_relax_fused_skip": (ctypes.c_int, [ctypes.POINTER(FwxSlab), ctypes.POINTER(FwxPivots),
                                                ctypes.POINTER(FwxFusedScratch),
                                                ctypes.POINTER(FwxTrace), c_vp, c_i32, c_i32, c_i32,
                                                c_vp]),
    "fwx_dev_check_nonneg": (ctypes.c_int, [ctypes.POINTER(FwxSlab), c_vp, c_vp]),
}

_LIB = None


def lib():
    """Load libfwx.so from the package directory.  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libfwx.so is missing at %s -- build it with `python -m floydwarshall_amd.build` "
                "(hipcc, gfx950).  floydwarshall_amd has no CPU fallback." % LIB_PATH)
        # One HIP runtime per process, and by default the one libfwx was BUILT AGAINST (/opt/rocm,
        # found through the library's RUNPATH): nothing else is imported here.  torch wheels bundle
        # their own libamdhip64.so.7 (same SONAME, an older ROCm); a program that needs torch in the
        # same process (floydwarshall_amd.dist: torch.distributed) must import torch BEFORE this
        # loader runs, so that libfwx's DT_NEEDED resolves to the copy torch initialises -- the other
        # order leaves torch with "No HIP GPUs are available".  That pairing works but is not the one
        # the library is built and fuzzed on; runtime_versions() reports it and a mismatch is logged
        # once (FWX_STRICT_RUNTIME=1 turns it into an error).
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.fwx_abi_version() != FWX_ABI_VERSION:
            raise RuntimeError("libfwx ABI version mismatch")
        _LIB = L
        _check_runtime(L)
    return _LIB


def runtime_versions():
    """(HIP_VERSION libfwx was compiled against, version of the HIP runtime it is bound to in this
    process, major.minor agree).  The runtime version is 0 where hipRuntimeGetVersion fails."""
    built, run = c_i32(0), c_i32(0)
    same = lib().fwx_hip_versions(ctypes.byref(built), ctypes.byref(run))
    return int(built.value), int(run.value), bool(same)


def _check_runtime(L):
    built, run = c_i32(0), c_i32(0)
    if L.fwx_hip_versions(ctypes.byref(built), ctypes.byref(run)) or run.value == 0:
        return
    msg = ("libfwx was built against HIP %d but is bound to HIP runtime %d in this process (another "
           "library -- a torch wheel bundles its own libamdhip64 -- was loaded first).  Load "
           "floydwarshall_amd before it, or do not import it at all, to run on the runtime the library "
           "was built and fuzzed on." % (built.value, run.value))
    if os.environ.get("FWX_STRICT_RUNTIME") == "1":
        raise RuntimeError(msg)
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


def check(status, what):
    if status < 0:
        raise FwxError(status, what)
    return status
