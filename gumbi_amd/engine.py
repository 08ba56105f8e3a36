"""ctypes binding of ``libgumbi_hip.so`` (C ABI in ``include/gumbi_hip.h``).

This is the only door between the Python front end and the numerics: there is no Python / numpy
implementation of the covariance build, the Cholesky factorisation or the posterior solves in
this package.  If the shared library is missing or no HIP device is present every call raises --
the product path never falls back to the CPU.
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

__all__ = ["Engine", "KernelSpec", "GumbiHipError", "library_path", "load_library", "KERNEL_KINDS",
           "ls_limits", "dist_plan", "GmbComm", "DistStep"]

GMB_MAX_DIMS, GMB_MAX_LIN, GMB_MAX_COREG, GMB_MAX_LEVELS = 16, 8, 4, 32
GMB_OK, GMB_EINVAL, GMB_ENOMEM, GMB_EHIP, GMB_ENOTPD, GMB_ENODEVICE = 0, -1, -2, -3, -4, -5
GMB_HOST, GMB_DEVICE = 0, 1

# pm.gp.cov names accepted by PymcGP._construct_kernels (gumbi/regression/pymc/GP.py:664-674)
KERNEL_KINDS = {"ExpQuad": 0, "Matern52": 1, "Matern32": 2, "Matern12": 3, "Exponential": 4}


class GumbiHipError(RuntimeError):
    """A libgumbi_hip call failed (bad HIP state, missing device, out of memory ...)."""


class _Spec(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("ard", C.c_int32),
        ("n_cont", C.c_int32),
        ("idx_cont", C.c_int32 * GMB_MAX_DIMS),
        ("n_lin", C.c_int32),
        ("idx_lin", C.c_int32 * GMB_MAX_LIN),
        ("n_coreg", C.c_int32),
        ("coreg_col", C.c_int32 * GMB_MAX_COREG),
        ("coreg_levels", C.c_int32 * GMB_MAX_COREG),
        ("out_col", C.c_int32),
        ("n_out", C.c_int32),
        ("hetero_noise", C.c_int32),
        ("additive", C.c_int32),
        ("jitter", C.c_double),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("kbuild_ms", C.c_double),
        ("chol_ms", C.c_double),
        ("chol_gemm_ms", C.c_double),
        ("chol_gemm_flops", C.c_double),
        ("chol_gemm_launches", C.c_int64),
        ("chol_leaf_ms", C.c_double),
        ("chol_trsm_ms", C.c_double),
        ("predict_ms", C.c_double),
        ("predict_gemm_ms", C.c_double),
        ("predict_gemm_flops", C.c_double),
        ("predict_gemm_launches", C.c_int64),
        ("grad_ms", C.c_double),
        ("grad_gemm_ms", C.c_double),
        ("grad_gemm_flops", C.c_double),
        ("kbuild_bytes", C.c_double),
        ("total_gemm_ms", C.c_double),
        ("total_gemm_flops", C.c_double),
        ("total_gemm_launches", C.c_int64),
        ("total_kbuild_ms", C.c_double),
        ("total_kbuild_bytes", C.c_double),
        ("total_kbuild_launches", C.c_int64),
        ("masked_gemm_ms", C.c_double),
        ("masked_gemm_flops", C.c_double),
        ("masked_cus", C.c_int64),
        ("total_gemm_wall_ms", C.c_double),
        ("total_chol_gemm_ms", C.c_double),
        ("total_chol_gemm_flops", C.c_double),
        ("total_chol_gemm_launches", C.c_int64),
        ("total_chol_gemm_wall_ms", C.c_double),
        ("total_chol_panel_gemm_ms", C.c_double),
        ("total_chol_panel_gemm_flops", C.c_double),
        # multi-GPU driver: communication probes of the last gmb_dist_factorize / gmb_dist_nlml (this rank)
        ("dist_world", C.c_int64),
        ("dist_chol_collectives", C.c_int64),
        ("dist_chol_comm_bytes", C.c_double),
        ("dist_chol_comm_ms", C.c_double),
        ("dist_chol_comm_exposed_ms", C.c_double),
        ("dist_chol_main_wait_ms", C.c_double),
        ("dist_chol_bulk_wait_ms", C.c_double),
        ("dist_grad_collectives", C.c_int64),
        ("dist_grad_comm_bytes", C.c_double),
        ("dist_grad_comm_ms", C.c_double),
        ("dist_grad_comm_exposed_ms", C.c_double),
        ("dist_lockstep_repairs", C.c_int64),
        ("total_chol_tile_ms", C.c_double),
        ("total_chol_tile_flops", C.c_double),
        ("total_chol_tile_launches", C.c_int64),
        ("total_eval_tile_ms", C.c_double),
        ("total_eval_tile_flops", C.c_double),
        ("total_eval_tile_launches", C.c_int64),
        # ABI 10: every launch issued AS a Cholesky trailing update (the population of ``total_chol_gemm_*`` up to round 4), and
        # which form the last prediction took
        ("total_chol_update_all_ms", C.c_double),
        ("total_chol_update_all_flops", C.c_double),
        ("total_chol_update_all_launches", C.c_int64),
        ("predict_gemm_form", C.c_int64),
        ("total_chol_panel_tile_ms", C.c_double),
        ("total_chol_panel_tile_flops", C.c_double),
        ("total_chol_panel_tile_launches", C.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


@dataclass
class KernelSpec:
    """Which columns of X feed which covariance term (mirror of ``gmb_kernel_spec``)."""

    D: int
    idx_cont: list
    kind: str | int = "ExpQuad"
    ard: bool = True
    idx_lin: list = field(default_factory=list)
    coreg: list = field(default_factory=list)  # [(column, n_levels), ...]
    out_col: int = -1
    n_out: int = 0
    hetero_noise: bool = True
    jitter: float = 1e-6
    additive: bool = False  # a global kernel + one kernel per coregion dim (specify_model(additive=True))

    @property
    def kind_id(self) -> int:
        if isinstance(self.kind, str):
            if self.kind not in KERNEL_KINDS:
                raise ValueError(f"Continuous kernel must be one of {list(KERNEL_KINDS)}, got {self.kind!r}")
            return KERNEL_KINDS[self.kind]
        return int(self.kind)

    def as_dict(self) -> dict:
        return dict(D=int(self.D), kind=self.kind_id, ard=bool(self.ard),
                    idx_cont=[int(i) for i in self.idx_cont], idx_lin=[int(i) for i in self.idx_lin],
                    coreg=[(int(c), int(n)) for c, n in self.coreg], out_col=int(self.out_col),
                    n_out=int(self.n_out), hetero_noise=bool(self.hetero_noise), jitter=float(self.jitter),
                    additive=bool(self.additive))

    def theta_size(self) -> int:
        n = (len(self.idx_cont) if self.ard else 1) + 2
        if self.idx_lin:
            n += len(self.idx_lin) + 1
        n += sum(3 * L for _, L in self.coreg)
        if self.out_col >= 0:
            n += 3 * self.n_out * (2 if self.hetero_noise else 1)
        if self.additive:  # one more [ls | eta | (c, tau)] block per coregion dim
            blk = (len(self.idx_cont) if self.ard else 1) + 1 + ((len(self.idx_lin) + 1) if self.idx_lin else 0)
            n += len(self.coreg) * blk
        return n

    def to_c(self) -> _Spec:
        if not 1 <= len(self.idx_cont) <= GMB_MAX_DIMS:
            raise ValueError(f"between 1 and {GMB_MAX_DIMS} continuous dimensions are supported")
        if len(self.idx_lin) > GMB_MAX_LIN:
            raise ValueError(f"at most {GMB_MAX_LIN} linear dimensions are supported")
        if len(self.coreg) > GMB_MAX_COREG:
            raise ValueError(f"at most {GMB_MAX_COREG} categorical dimensions are supported")
        s = _Spec()
        s.kind, s.ard = self.kind_id, int(bool(self.ard))
        s.n_cont = len(self.idx_cont)
        for i, v in enumerate(self.idx_cont):
            s.idx_cont[i] = int(v)
        s.n_lin = len(self.idx_lin)
        for i, v in enumerate(self.idx_lin):
            s.idx_lin[i] = int(v)
        s.n_coreg = len(self.coreg)
        for i, (col, lev) in enumerate(self.coreg):
            s.coreg_col[i], s.coreg_levels[i] = int(col), int(lev)
        s.out_col, s.n_out = int(self.out_col), int(self.n_out)
        s.hetero_noise = int(bool(self.hetero_noise))
        s.additive = int(bool(self.additive))
        s.jitter = float(self.jitter)
        return s


# ------------------------------------------------------------------------------------------------
_LIB = None
_DBL_P = C.POINTER(C.c_double)


def library_path() -> Path:
    override = os.environ.get("GUMBI_HIP_LIB")
    if override:
        return Path(override)
    return Path(__file__).resolve().parent / "lib" / "libgumbi_hip.so"


class GmbComm(C.Structure):
    """``gmb_comm``: the one collective the multi-GPU driver needs (include/gumbi_hip.h)."""

    ALL_GATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("ctx", C.c_void_p), ("all_gather", ALL_GATHER)]


class DistStep(C.Structure):
    """``gmb_dist_step``: one step of a rank's schedule (include/gumbi_hip.h)."""

    _fields_ = [(n, C.c_int32) for n in ("op", "c0", "c1", "lo", "hi", "first", "count", "maxcount", "stream")] + [
        ("elems", C.c_int64)]
    OPS = ("KBUILD", "SQUARE", "PANEL", "UPDATE", "FORK", "JOIN", "SOLVE", "TAIL")

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_}
        d["op"] = self.OPS[d["op"]]
        return d


_SIGNATURES = {
    "gmb_abi_version": (C.c_int, []),
    "gmb_device_count": (C.c_int, []),
    "gmb_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "gmb_create_sibling": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "gmb_destroy": (None, [C.c_void_p]),
    "gmb_last_error": (C.c_char_p, [C.c_void_p]),
    "gmb_stream": (C.c_void_p, [C.c_void_p]),
    "gmb_set_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int32]),
    "gmb_set_y": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "gmb_set_kernel": (C.c_int, [C.c_void_p, C.POINTER(_Spec)]),
    "gmb_theta_size": (C.c_int, [C.POINTER(_Spec)]),
    "gmb_set_theta": (C.c_int, [C.c_void_p, _DBL_P, C.c_int32]),
    "gmb_factorize": (C.c_int, [C.c_void_p]),
    "gmb_notpd_index": (C.c_int64, [C.c_void_p]),
    "gmb_factor_valid": (C.c_int, [C.c_void_p]),
    "gmb_nlml": (C.c_int, [C.c_void_p, _DBL_P, _DBL_P]),
    "gmb_evaluate": (C.c_int, [C.c_void_p, _DBL_P, C.c_int32, _DBL_P, _DBL_P]),
    "gmb_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                              C.c_void_p, C.c_int32]),
    "gmb_ls_limits": (C.c_int, [C.c_int32, _DBL_P, C.c_int64, C.c_int32, C.c_int64, C.c_int32, _DBL_P, _DBL_P]),
    "gmb_mfma_f64_peak": (C.c_int, [C.c_int32, _DBL_P, _DBL_P]),
    "gmb_mfma_f64_sustained": (C.c_int, [C.c_int32, C.c_double, _DBL_P, _DBL_P, _DBL_P, C.POINTER(C.c_int64)]),
    "gmb_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "gmb_timings_get": (C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    "gmb_copy_factor": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _DBL_P]),
    "gmb_copy_v": (C.c_int, [C.c_void_p, _DBL_P]),
    "gmb_blk_potrf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gmb_copy_alpha": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "gmb_rccl_unique_id": (C.c_int, [C.c_char_p, C.c_void_p]),
    "gmb_rccl_comm_create": (C.c_int, [C.POINTER(C.POINTER(GmbComm)), C.c_char_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32]),
    "gmb_rccl_comm_destroy": (None, [C.POINTER(GmbComm)]),
    "gmb_rccl_comm_ranks": (C.c_int, [C.POINTER(GmbComm)]),
    "gmb_rccl_comm_split": (C.c_int, [C.POINTER(GmbComm)]),
    "gmb_rccl_last_error": (C.c_char_p, []),
    "gmb_dist_plan": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(DistStep), C.c_int64]),
    "gmb_dist_factorize": (C.c_int, [C.c_void_p, C.POINTER(GmbComm), C.c_int32]),
    "gmb_dist_nlml": (C.c_int, [C.c_void_p, C.POINTER(GmbComm), _DBL_P, _DBL_P]),
    "gmb_dist_predict": (C.c_int, [C.c_void_p, C.POINTER(GmbComm), C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int32]),
    "gmb_debug_cov_grid": (C.c_int64, [C.c_int32] * 7 + [C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]),
    "gmb_debug_tile_list": (C.c_int64, [C.c_int32] * 12 + [C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int32)]),
    "gmb_debug_chol_task": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gmb_chol_task_trace": (C.c_int64, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "gmb_set_chol_scheme": (C.c_int, [C.c_void_p, C.c_int32]),
    "gmb_debug_assume_factored": (C.c_int, [C.c_void_p]),
    "gmb_set_eval_pairs": (C.c_int, [C.c_void_p, C.c_int32]),
    "gmb_set_predict_form": (C.c_int, [C.c_void_p, C.c_int32]),
    "gmb_reserve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "gmb_set_grad_scheme": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "gmb_dist_set_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "gmb_resident_bytes": (C.c_int64, [C.c_void_p, C.c_int32]),
    "gmb_debug_eval_tasks": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.c_int64]),
    "gmb_debug_chol_lose_tickets": (C.c_int, [C.c_void_p, C.c_int32]),
    "gmb_blk_invert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "gmb_blk_covariance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "gmb_blk_trsm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                               C.c_int32]),
    "gmb_blk_gemm_nt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                  C.c_int32, C.c_int64]),
}


def exported_symbols() -> list:
    """Every entry point ``include/gumbi_hip.h`` declares."""
    return list(_SIGNATURES)


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so with the
    same SONAME (libamdhip64.so.7) as /opt/rocm's; whichever is mapped first serves every later
    request for that SONAME, but torch asks for its copy by *file name* and would map a second
    runtime if ours came from /opt/rocm first.  Mapping torch's copy up front (when torch is
    installed) makes libgumbi_hip.so, torch and RCCL share one runtime -- and therefore streams,
    events and device pointers -- in either import order."""
    if os.environ.get("GUMBI_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # fall back to whatever libamdhip64.so.7 the loader finds


#: GMB_ABI_VERSION of include/gumbi_hip.h this binding was written against (the layout of
#: ``gmb_kernel_spec`` changed with version 2: ``additive``; ``gmb_timings`` grew with version 3; version 4 replaced the block-level
#: multi-GPU entry points by the native driver ``gmb_dist_*``; version 5 added ``gmb_blk_covariance``, ``gmb_set_y`` and ``gmb_create_sibling``;
#: version 6 the communication fields of ``gmb_timings`` and ``gmb_rccl_comm_ranks``; version 7 ``gmb_evaluate``, the tile
#: Cholesky's doors ``gmb_set_chol_scheme`` / ``gmb_chol_task_trace`` / ``gmb_debug_chol_*``; version 8 the persistent evaluation
#: launch: ``gmb_set_grad_scheme``, ``gmb_debug_eval_tasks``, ``total_eval_tile_*`` in ``gmb_timings``; version 9 the timing
#: tools' door ``gmb_debug_assume_factored``, the guards between single-engine and capacity-mode factorisations, ``gmb_set_eval_pairs``;
#: version 10 ``gmb_set_predict_form`` (GEMM-form prediction behind a fit), ``total_chol_update_all_*`` / ``predict_gemm_form`` in
#: ``gmb_timings``, ``gmb_debug_assume_factored`` behind GUMBI_HIP_DEBUG_DOORS=1)
ABI_VERSION = 10


def load_library():
    """Load libgumbi_hip.so (once) and attach the prototypes.  Raises if it is not built or was
    built from another version of the header."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise GumbiHipError(
            f"{path} not found: the HIP engine is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C gumbi_amd/csrc`). gumbi_amd has no CPU fallback."
        )
    _preload_hip_runtime()
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = restype, argtypes
    if lib.gmb_abi_version() != ABI_VERSION:
        raise GumbiHipError(f"{path} has ABI version {lib.gmb_abi_version()}, this binding needs {ABI_VERSION}: "
                            f"rebuild it (`python -m gumbi_amd.build`)")
    _LIB = lib
    return lib


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p)


def _dptr(arr):
    return arr.ctypes.data_as(_DBL_P)


def device_count() -> int:
    n = load_library().gmb_device_count()
    return max(n, 0)


class Engine:
    """One GP resident on one MI355X: data, covariance / factor, predict workspaces."""

    #: the evaluations do not run on the host's BLAS (HipGP.find_MAP keeps OpenBLAS to one thread inside the optimiser's loop)
    host_blas_free = True

    def __init__(self, device: int = 0, stream: int | None = None, sibling_of: "Engine | None" = None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._peer = sibling_of  # a sibling borrows the peer's HIP streams: the peer must outlive it
        if sibling_of is not None:  # gmb_create_sibling; close the sibling first
            rc = self._lib.gmb_create_sibling(C.byref(self._h), sibling_of._h)
        else:
            rc = self._lib.gmb_create(C.byref(self._h), int(device), C.c_void_p(stream) if stream else None)
        if rc == GMB_ENODEVICE:
            raise GumbiHipError("no HIP device visible: gumbi_amd needs an MI355X (gfx950); there is no CPU fallback")
        if rc != GMB_OK:
            raise GumbiHipError(f"gmb_create failed with status {rc}")
        self.spec = None
        self.N = 0
        self.D = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gmb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- error mapping (Regressor error surface: ValueError / LinAlgError) ---------
    def _check(self, rc, what):
        if rc == GMB_OK:
            return
        msg = self._lib.gmb_last_error(self._h)
        msg = msg.decode() if msg else ""
        if rc == GMB_EINVAL:
            raise ValueError(f"{what}: {msg}")
        if rc == GMB_ENOTPD:
            raise np.linalg.LinAlgError(f"{what}: {msg}")
        if rc == GMB_ENOMEM:
            raise MemoryError(f"{what}: {msg}")
        if rc == GMB_ENODEVICE:
            raise GumbiHipError(f"{what}: no HIP device (no CPU fallback exists)")
        raise GumbiHipError(f"{what}: status {rc}: {msg}")

    @property
    def stream(self) -> int:
        return int(self._lib.gmb_stream(self._h) or 0)

    # -- model definition -----------------------------------------------------------------------
    def set_data(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if X.ndim != 2 or y.ndim != 1 or X.shape[0] != y.shape[0]:
            raise ValueError(f"X must be (N, D) and y (N,), got {X.shape} and {y.shape}")
        self._check(self._lib.gmb_set_data(self._h, _ptr(X), X.shape[0], X.shape[1], X.shape[1], _ptr(y), GMB_HOST),
                    "gmb_set_data")
        self.N, self.D = X.shape

    def set_y(self, y):
        """New observations for the same inputs: everything but the factorisation stays."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.shape != (self.N,):
            raise ValueError(f"y must be ({self.N},), got {y.shape}")
        self._check(self._lib.gmb_set_y(self._h, _ptr(y), GMB_HOST), "gmb_set_y")

    def set_data_device(self, x_ptr: int, N: int, D: int, ldx: int, y_ptr: int):
        """Inputs already in HBM (e.g. ``torch.Tensor.data_ptr()``)."""
        self._check(self._lib.gmb_set_data(self._h, C.c_void_p(x_ptr), N, D, ldx, C.c_void_p(y_ptr), GMB_DEVICE),
                    "gmb_set_data")
        self.N, self.D = N, D

    def set_kernel(self, spec: KernelSpec):
        cs = spec.to_c()
        self._check(self._lib.gmb_set_kernel(self._h, C.byref(cs)), "gmb_set_kernel")
        self.spec = spec

    def set_theta(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        self._check(self._lib.gmb_set_theta(self._h, _dptr(theta), theta.size), "gmb_set_theta")

    # -- hot path ----------------------------------------------------------------------------------
    def factorize(self):
        self._check(self._lib.gmb_factorize(self._h), "gmb_factorize")

    def notpd_index(self) -> int:
        return int(self._lib.gmb_notpd_index(self._h))

    def factor_is_current(self) -> bool:
        """True while a usable factorisation of the current (data, kernel, theta) is resident -- also after
        ``nlml(grad=True)``: ``predict`` may follow without another ``factorize``."""
        return bool(self._lib.gmb_factor_valid(self._h))

    def nlml(self, grad: bool = False):
        val = C.c_double()
        if not grad:
            self._check(self._lib.gmb_nlml(self._h, C.byref(val), None), "gmb_nlml")
            return val.value
        g = np.empty(self.spec.theta_size(), dtype=np.float64)
        self._check(self._lib.gmb_nlml(self._h, C.byref(val), _dptr(g)), "gmb_nlml")
        return val.value, g

    def evaluate(self, theta, grad: bool = True):
        """``set_theta`` + ``factorize`` + ``nlml`` in one call (no host round trip between the factorisation and the
        gradient): ``(nlml, gradient)`` or ``nlml``."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        val = C.c_double()
        g = np.empty(self.spec.theta_size(), dtype=np.float64) if grad else None
        self._check(self._lib.gmb_evaluate(self._h, _dptr(theta), theta.size, C.byref(val), _dptr(g) if grad else None), "gmb_evaluate")
        return (val.value, g) if grad else val.value

    def predict(self, Xs, with_noise=True):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        if Xs.ndim != 2 or Xs.shape[1] != self.D:
            raise ValueError(f"points_array must be (M, {self.D}), got {Xs.shape}")
        M = Xs.shape[0]
        mean = np.empty(M, dtype=np.float64)
        var = np.empty(M, dtype=np.float64)
        self._check(
            self._lib.gmb_predict(self._h, _ptr(Xs), M, Xs.shape[1], int(bool(with_noise)), _ptr(mean), _ptr(var),
                                  GMB_HOST),
            "gmb_predict",
        )
        return mean, var

    def predict_device(self, xs_ptr: int, M: int, ldxs: int, mean_ptr: int, var_ptr: int, with_noise=True):
        self._check(
            self._lib.gmb_predict(self._h, C.c_void_p(xs_ptr), M, ldxs, int(bool(with_noise)), C.c_void_p(mean_ptr),
                                  C.c_void_p(var_ptr), GMB_DEVICE),
            "gmb_predict",
        )

    # -- introspection -----------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self._lib.gmb_set_profiling(self._h, int(bool(on))), "gmb_set_profiling")

    def timings(self) -> dict:
        t = Timings()
        self._check(self._lib.gmb_timings_get(self._h, C.byref(t)), "gmb_timings_get")
        return t.as_dict()

    #: schedules of the Cholesky (``set_chol_scheme``): by size / plain recursion / masked look-ahead / persistent tile kernel
    CHOL_BY_SIZE, CHOL_RECURSION, CHOL_LOOKAHEAD, CHOL_TILES, CHOL_RECURSION_TILE_PANELS = -1, 0, 2, 3, 4

    def set_chol_scheme(self, scheme: int) -> int:
        """Schedule of the following factorisations; returns the previous setting."""
        prev = int(self._lib.gmb_set_chol_scheme(self._h, int(scheme)))  # previous + 1, or a negative status
        if prev < 0:
            self._check(prev, "gmb_set_chol_scheme")
        return prev - 1

    #: schedules of the gradient's inverse / Sigma^-1 (``set_grad_scheme``)
    GRAD_BY_SIZE, GRAD_LAUNCH_TREE, GRAD_TILES, GRAD_FUSED = -1, 0, 1, 2

    def set_grad_scheme(self, scheme: int, lag: int = -1) -> int:
        """Schedule of the gradient's inverse and Sigma^-1 for the following evaluations (``lag``: block columns the inverse
        follows the factorisation by in the fused launch's ticket order); returns the previous setting."""
        prev = int(self._lib.gmb_set_grad_scheme(self._h, int(scheme), int(lag)))
        if prev < 0:
            self._check(prev, "gmb_set_grad_scheme")
        return prev - 1

    def debug_lose_tickets(self, n: int):
        """Fault injection (tests): the next tile factorisation never computes the tiles of tickets 0 .. n-1."""
        self._check(self._lib.gmb_debug_chol_lose_tickets(self._h, int(n)), "gmb_debug_chol_lose_tickets")

    def chol_task_trace(self, enable: int = -1):
        """Per-task stamps of the last persistent tile factorisation: ``(tiles, stamps)`` with ``tiles`` the (I, J) of every
        ticket and ``stamps`` (ntasks, 4) seconds since the first ticket (taken / contraction done / solve input ready /
        published); ``None`` when the last factorisation recorded nothing.  ``enable`` = 1 / 0 switches the stamps of the
        FOLLOWING factorisations on / off."""
        n = int(self._lib.gmb_chol_task_trace(self._h, int(enable), None, 0))
        if n <= 0:
            return None
        raw = np.zeros((n, 4), dtype=np.uint64)
        self._check(min(0, int(self._lib.gmb_chol_task_trace(self._h, -1, _ptr(raw), n))), "gmb_chol_task_trace")
        nct = (self.N + 127) // 128
        nrt = (self.N + 1 + 127) // 128
        tiles = np.array([chol_task(t, nct, nrt) for t in range(n)], dtype=np.int32)
        t0 = raw[raw > 0].min() if (raw > 0).any() else 0
        return tiles, (raw.astype(np.int64) - int(t0)) * 1e-8

    def eval_task_trace(self, with_chol: bool = True, lag: int = -1):
        """Per-task stamps of the last persistent EVALUATION launch (csrc/eval_tiles.hpp; switch them on with
        ``chol_task_trace(1)``): ``(tasks, stamps)`` with ``tasks`` an (ntasks, 3) array of (kind, I, J) in ticket order and
        ``stamps`` as in ``chol_task_trace``; ``None`` when nothing was recorded or the task count does not match."""
        n = int(self._lib.gmb_chol_task_trace(self._h, -1, None, 0))
        if n <= 0:
            return None
        nct = (self.N + 127) // 128
        nrt = (self.N + 1 + 127) // 128
        if lag < 0:
            lag = max(2, min(24, nct // 4))  # the engine's default (csrc/engine.hip: eval_tiles)
        tasks = np.array(eval_task_list(nct, nrt, with_chol, lag), dtype=np.int32)
        if len(tasks) != n:
            return None
        raw = np.zeros((n, 4), dtype=np.uint64)
        self._check(min(0, int(self._lib.gmb_chol_task_trace(self._h, -1, _ptr(raw), n))), "gmb_chol_task_trace")
        t0 = raw[raw > 0].min() if (raw > 0).any() else 0
        return tasks, (raw.astype(np.int64) - int(t0)) * 1e-8

    def copy_factor(self, r0=0, nr=None, c0=0, nc=None):
        nr = self.N if nr is None else nr
        nc = self.N if nc is None else nc
        out = np.empty((nr, nc), dtype=np.float64)
        self._check(self._lib.gmb_copy_factor(self._h, r0, nr, c0, nc, _dptr(out)), "gmb_copy_factor")
        return out

    def copy_v(self):
        out = np.empty(self.N, dtype=np.float64)
        self._check(self._lib.gmb_copy_v(self._h, _dptr(out)), "gmb_copy_v")
        return out

    def copy_alpha(self):
        """alpha = Sigma^-1 y (available after ``nlml(grad=True)``)."""
        out = np.empty(self.N, dtype=np.float64)
        self._check(self._lib.gmb_copy_alpha(self._h, _dptr(out)), "gmb_copy_alpha")
        return out

    # -- ONE GP over several GPUs (native driver, gumbi_amd/csrc/dist_driver.hpp) -----------------------------
    #: how the ranks of ``dist_*`` hold the factor (``set_dist_mode``): every rank all of it / owned block rows + panel buffers
    DIST_REPLICATED, DIST_CAPACITY = 0, 1

    def set_eval_pairs(self, mode: int) -> int:
        """Sigma^-1 tile tasks in pairs (128 x 256): -1 by size, 0 never, 1 always (same bits either way); returns the previous mode."""
        prev = int(self._lib.gmb_set_eval_pairs(self._h, int(mode)))
        if prev < 0:
            self._check(prev, "gmb_set_eval_pairs")
        return prev - 1

    def set_predict_form(self, form: int) -> int:
        """How ``predict`` forms K(X*, X) L^-T: -1 / 1 = one GEMM against the inverse factor whenever the fit's last gradient
        evaluation left it resident (tile path), 0 = always the triangular solve; returns the previous mode."""
        prev = int(self._lib.gmb_set_predict_form(self._h, int(form)))
        if prev < 0:
            self._check(prev, "gmb_set_predict_form")
        return prev - 1

    def reserve(self, gradient: bool = True, M: int = 0):
        """Allocate now what the following factorisation / gradient / prediction of ``M`` points would allocate on first use."""
        self._check(self._lib.gmb_reserve(self._h, int(bool(gradient)), int(M)), "gmb_reserve")

    def debug_assume_factored(self):
        """Timing tools only: the last factorisation counts as valid whatever it produced (``gmb_debug_assume_factored``)."""
        self._check(self._lib.gmb_debug_assume_factored(self._h), "gmb_debug_assume_factored")

    def set_dist_mode(self, mode: int) -> int:
        """Replicated (0) or capacity (1) mode of the multi-GPU driver for the following ``dist_factorize``; returns the previous mode."""
        prev = int(self._lib.gmb_dist_set_mode(self._h, int(mode)))
        if prev < 0:
            self._check(prev, "gmb_dist_set_mode")
        return prev

    def resident_bytes(self, peak: bool = False) -> int:
        """Device bytes the engine holds through its own allocations (``peak``: the largest value so far)."""
        return int(self._lib.gmb_resident_bytes(self._h, int(bool(peak))))

    def dist_factorize(self, comm, panel_blocks: int = 0):
        """Collective: ``gmb_factorize`` over the ranks of ``comm`` (a :class:`gumbi_amd.distributed` comm)."""
        self._check(self._lib.gmb_dist_factorize(self._h, comm.handle, int(panel_blocks)), "gmb_dist_factorize")

    def dist_nlml(self, comm, grad: bool = False):
        val = C.c_double()
        if not grad:
            self._check(self._lib.gmb_dist_nlml(self._h, comm.handle, C.byref(val), None), "gmb_dist_nlml")
            return val.value
        g = np.empty(self.spec.theta_size(), dtype=np.float64)
        self._check(self._lib.gmb_dist_nlml(self._h, comm.handle, C.byref(val), _dptr(g)), "gmb_dist_nlml")
        return val.value, g

    def dist_predict(self, comm, Xs, with_noise=True):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        if Xs.ndim != 2 or Xs.shape[1] != self.D:
            raise ValueError(f"points_array must be (M, {self.D}), got {Xs.shape}")
        M = Xs.shape[0]
        mean = np.empty(M, dtype=np.float64)
        var = np.empty(M, dtype=np.float64)
        self._check(self._lib.gmb_dist_predict(self._h, comm.handle, _ptr(Xs), M, Xs.shape[1], int(bool(with_noise)),
                                               _ptr(mean), _ptr(var), GMB_HOST), "gmb_dist_predict")
        return mean, var

    # -- block-level operations (device pointers) --------------------------------------------------------
    def blk_potrf(self, a_ptr, lda, nvalid, dinv16_ptr, logdet_ptr=0, info_ptr=0):
        """Factor a 128 x 128 block in place; ``dinv16_ptr`` (8 x 256 doubles) receives the inverses of
        its 16 x 16 diagonal sub-blocks, the operand of :meth:`blk_trsm` / :meth:`blk_invert`."""
        self._check(self._lib.gmb_blk_potrf(self._h, C.c_void_p(a_ptr), lda, nvalid,
                                            C.c_void_p(dinv16_ptr) if dinv16_ptr else None,
                                            C.c_void_p(logdet_ptr) if logdet_ptr else None,
                                            C.c_void_p(info_ptr) if info_ptr else None), "gmb_blk_potrf")

    def blk_invert(self, l_ptr, lda, nvalid, dinv16_ptr, inv_ptr):
        self._check(self._lib.gmb_blk_invert(self._h, C.c_void_p(l_ptr), lda, nvalid, C.c_void_p(dinv16_ptr),
                                             C.c_void_p(inv_ptr)), "gmb_blk_invert")

    def blk_covariance(self, out_ptr, ldo):
        """The covariance build alone into a caller's device buffer (lower-triangle tiles, y row, padding)."""
        self._check(self._lib.gmb_blk_covariance(self._h, C.c_void_p(out_ptr), ldo), "gmb_blk_covariance")

    def blk_trsm(self, b_ptr, ldb, nrows, l_ptr, ldl, dinv16_ptr, nvalid=128):
        """B <- B inv(L)^T in place on ``nrows`` (multiple of 16) rows."""
        self._check(self._lib.gmb_blk_trsm(self._h, C.c_void_p(b_ptr), ldb, nrows, C.c_void_p(l_ptr), ldl,
                                           C.c_void_p(dinv16_ptr), nvalid), "gmb_blk_trsm")

    def blk_gemm_nt(self, c_ptr, ldc, a_ptr, lda, b_ptr, ldb, m, n, k, alpha, beta, tri=0, tri_shift=0):
        self._check(self._lib.gmb_blk_gemm_nt(self._h, C.c_void_p(c_ptr), ldc, C.c_void_p(a_ptr), lda,
                                              C.c_void_p(b_ptr), ldb, m, n, k, alpha, beta, tri, tri_shift),
                    "gmb_blk_gemm_nt")


def chol_task(t: int, nct: int, nrt: int):
    """Ticket ``t`` of the persistent tile Cholesky -> its tile (I, J) (host-side enumeration shared with the kernel)."""
    lib = load_library()
    i, j = C.c_int32(), C.c_int32()
    n = lib.gmb_debug_chol_task(int(t), int(nct), int(nrt), C.byref(i), C.byref(j))
    if n < 0:
        raise ValueError("gmb_debug_chol_task: bad arguments")
    return i.value, j.value


def eval_task_list(nct: int, nrt: int, with_chol: bool = True, lag: int = 1, pairs: bool = False) -> list:
    """[(kind, I, J), ...] of the persistent evaluation launch in ticket order (host-only; kind 0 = Cholesky tile,
    1 = tile of U = L^-T, 2 = tile of Sigma^-1, 3 = alpha (and v) of block row I).  ``pairs``: the Sigma^-1 tasks as the launch
    lists them for large matrices -- ``(2, I, J + 0x4000)`` stands for the two tiles (I, J) and (I, J + 1)."""
    lib = load_library()
    mode = int(bool(with_chol)) | (2 if pairs else 0)
    n = lib.gmb_debug_eval_tasks(int(nct), int(nrt), mode, int(lag), None, 0)
    if n < 0:
        raise ValueError("gmb_debug_eval_tasks: bad arguments")
    buf = (C.c_uint32 * n)()
    lib.gmb_debug_eval_tasks(int(nct), int(nrt), mode, int(lag), buf, n)
    return [(w >> 30, (w >> 15) & 0x7FFF, w & 0x7FFF) for w in buf]


def chol_task_count(nct: int, nrt: int) -> int:
    return int(load_library().gmb_debug_chol_task(-1, int(nct), int(nrt), None, None))


def gemm_tile_list(mt, nt, bm=128, bn=128, k=1024, tri=0, tri_off=0, nblk_stride=1, klo_n=0, khi_n=0, order=0, strip=0):
    """(grid, [(block, tm, tn), ...]) of one GEMM launch as the kernel enumerates it (host-only)."""
    lib = load_library()
    args = [int(v) for v in (mt, nt, bm, bn, k, tri, tri_off, nblk_stride, klo_n, khi_n, order, strip)]
    grid = C.c_int32()
    n = lib.gmb_debug_tile_list(*args, None, 0, C.byref(grid))
    if n < 0:
        raise ValueError("gmb_debug_tile_list: bad arguments")
    buf = (C.c_int32 * (3 * n))()
    lib.gmb_debug_tile_list(*args, buf, n, C.byref(grid))
    arr = np.frombuffer(buf, dtype=np.int32).reshape(n, 3)
    return int(grid.value), [tuple(int(v) for v in row) for row in arr]


def cov_grid(ti, tj, strip=1, tri_grid=1, row_first=0, row_stride=0, keep_order=1):
    """(grid, array of (block, tile row, tile column)) of one covariance-build launch as cov_tile_kernel enumerates
    it (host-only)."""
    lib = load_library()
    args = [int(v) for v in (ti, tj, strip, tri_grid, row_first, row_stride, keep_order)]
    grid = C.c_int64()
    n = lib.gmb_debug_cov_grid(*args, None, 0, C.byref(grid))
    if n < 0:
        raise ValueError("gmb_debug_cov_grid: bad arguments")
    buf = (C.c_int32 * max(3 * n, 1))()
    lib.gmb_debug_cov_grid(*args, buf, n, C.byref(grid))
    return int(grid.value), np.frombuffer(buf, dtype=np.int32)[: 3 * n].reshape(n, 3).copy()


def dist_plan(N: int, rank: int, world: int, panel_blocks: int = 0) -> list:
    """The schedule rank ``rank`` of ``world`` executes in ``gmb_dist_factorize`` (host-only; no device needed)."""
    lib = load_library()
    n = lib.gmb_dist_plan(int(N), int(rank), int(world), int(panel_blocks), None, 0)
    if n < 0:
        raise ValueError(f"gmb_dist_plan: bad arguments (N={N}, rank={rank}, world={world})")
    buf = (DistStep * n)()
    lib.gmb_dist_plan(int(N), int(rank), int(world), int(panel_blocks), buf, n)
    return [st.as_dict() for st in buf]


def ls_limits(X, ard: bool, device: int = 0):
    """Minimum non-zero and maximum pairwise distance per group (``gmb_ls_limits``)."""
    lib = load_library()
    X = np.ascontiguousarray(X, dtype=np.float64)
    n_groups = X.shape[1] if ard else 1
    lo = np.empty(n_groups)
    hi = np.empty(n_groups)
    rc = lib.gmb_ls_limits(int(device), _dptr(X), X.shape[0], X.shape[1], X.shape[1], int(bool(ard)), _dptr(lo),
                           _dptr(hi))
    if rc == GMB_ENODEVICE:
        raise GumbiHipError("gmb_ls_limits: no HIP device (no CPU fallback exists)")
    if rc != GMB_OK:
        raise GumbiHipError(f"gmb_ls_limits failed with status {rc}")
    return lo, hi


def mfma_f64_sustained(device: int = 0, seconds: float = 1.0) -> dict:
    """The register-only MFMA loop run back to back for ``seconds``: mean / worst launch rate (TFLOP/s), the
    shader clock it ran at (MHz) and the number of launches (``gmb_mfma_f64_sustained``)."""
    mean, worst, mhz, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    rc = load_library().gmb_mfma_f64_sustained(int(device), float(seconds), C.byref(mean), C.byref(worst), C.byref(mhz),
                                                C.byref(n))
    if rc != GMB_OK:
        raise GumbiHipError(f"gmb_mfma_f64_sustained failed with status {rc}")
    return {"tflops_mean": mean.value, "tflops_min": worst.value, "shader_mhz": mhz.value, "launches": int(n.value),
            "seconds": float(seconds)}


def mfma_f64_peak(device: int = 0):
    """(TFLOP/s, shader cycles per MFMA as seen by one wave) from ``gmb_mfma_f64_peak``."""
    out, cyc = C.c_double(), C.c_double()
    rc = load_library().gmb_mfma_f64_peak(int(device), C.byref(out), C.byref(cyc))
    if rc != GMB_OK:
        raise GumbiHipError(f"gmb_mfma_f64_peak failed with status {rc}")
    return out.value, cyc.value
