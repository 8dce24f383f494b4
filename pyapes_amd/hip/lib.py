"""ctypes binding of ``libpyapes_hip.so`` (C ABI: ``include/pyapes_hip.h``).

The library is built in-tree by ``pyapes_amd/csrc/build.sh`` (or
``__graft_entry__.build()``).  A missing library is a hard error -- there is no
fallback implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libpyapes_hip.so")

PA_F32, PA_F64 = 0, 1
BC_NONE, BC_DIRICHLET, BC_NEUMANN, BC_SYMMETRY, BC_PERIODIC = 0, 1, 2, 3, 4
BC_CODE = {"dirichlet": BC_DIRICHLET, "neumann": BC_NEUMANN, "symmetry": BC_SYMMETRY,
           "periodic": BC_PERIODIC}
OP_LAPLACIAN, OP_GRAD, OP_DIV_CENTRAL, OP_DIV_UPWIND_COMPAT, OP_DIV_UPWIND = 0, 1, 2, 3, 4
PA_COORD_XYZ, PA_COORD_RZ = 0, 1
PA_OK, PA_E_ARG, PA_E_HIP, PA_E_STATE, PA_E_NONFINITE = 0, -1, -2, -3, -4
PA_NSUM = 8


class PaError(RuntimeError):
    """Error reported by libpyapes_hip (code + text of pa_last_error)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libpyapes_hip error {code}: {msg}")
        self.code = code


class PaReport(C.Structure):
    _fields_ = [("itr", C.c_int64), ("tol", C.c_double), ("converge", C.c_int32),
                ("status", C.c_int32), ("rr", C.c_double), ("gpu_ms", C.c_double)]


class PaTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("has_coeff", C.c_int32), ("sign", C.c_double),
                ("coeff", C.c_double), ("coeff_field", C.c_void_p), ("u", C.c_double),
                ("u_field", C.c_void_p)]


class PaSlab(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "sums", "r_send_lo", "r_send_hi", "r_recv_lo", "r_recv_hi", "x_ghost_lo", "x_ghost_hi",
        "bc_far_lo0", "bc_far_lo1", "bc_far_hi0", "x_pack_lo1", "x_pack_hi0", "x_pack_hi1")]


class PaDivSpec(C.Structure):
    _fields_ = [("x", C.c_void_p * 3), ("u_int", C.c_void_p * 3), ("u_edge", C.c_void_p * 3),
                ("u", C.c_double), ("kind", C.c_int), ("edge", C.c_int)]


class PaExchange(C.Structure):
    _fields_ = [("nb_lo", C.c_int), ("nb_hi", C.c_int), ("send_lo", C.c_void_p), ("send_hi", C.c_void_p),
                ("recv_lo", C.c_void_p), ("recv_hi", C.c_void_p), ("n_send_lo", C.c_int64),
                ("n_send_hi", C.c_int64), ("n_recv_lo", C.c_int64), ("n_recv_hi", C.c_int64)]


# name -> (restype, argtypes); every symbol include/pyapes_hip.h declares
_I64P = C.POINTER(C.c_int64)
_F64P = C.POINTER(C.c_double)
_VP = C.c_void_p
SIGNATURES: dict[str, tuple[Any, list[Any]]] = {
    "pa_ctx_create": (C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    "pa_ctx_destroy": (C.c_int, [_VP]),
    "pa_ctx_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "pa_ctx_get_option": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int)]),
    "pa_ctx_set_stream": (C.c_int, [_VP, _VP]),
    "pa_cg_abort": (C.c_int, [_VP]),
    "pa_resident_plan": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int)]),
    "pa_resident_used": (C.c_int, [_VP]),
    "pa_last_error": (C.c_char_p, [_VP]),
    "pa_version": (C.c_char_p, []),
    "pa_grid_set": (C.c_int, [_VP, C.c_int, _I64P, _F64P, C.c_int, C.c_int64, C.c_int64]),
    "pa_bc_set": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_double, _VP, C.c_double]),
    "pa_bc_clear": (C.c_int, [_VP]),
    "pa_apply_bc": (C.c_int, [_VP, _VP]),
    "pa_eq_set": (C.c_int, [_VP, C.c_int, C.POINTER(PaTerm)]),
    "pa_aop": (C.c_int, [_VP, _VP, _VP, C.c_int]),
    "pa_rhs_adjust": (C.c_int, [_VP, _VP]),
    "pa_laplacian": (C.c_int, [_VP, _VP, _VP, C.c_int]),
    "pa_grad": (C.c_int, [_VP, _VP, _VP, C.c_int]),
    "pa_div": (C.c_int, [_VP, C.c_int, C.c_double, _VP, _VP, _VP]),
    "pa_div_edge": (C.c_int, [_VP, C.c_double, _VP, _VP, _VP]),
    "pa_coord_set": (C.c_int, [_VP, C.c_int, _VP]),
    "pa_div_general": (C.c_int, [_VP, C.POINTER(PaDivSpec), _VP]),
    "pa_diff_flux": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), _VP]),
    "pa_rfp_friction": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "pa_rfp_diffusion": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "pa_limiter": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, C.c_int64]),
    "pa_solver_keep_old": (C.c_int, [_VP, _VP]),
    "pa_cg": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64, C.POINTER(PaReport)]),
    "pa_bicgstab": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64, C.POINTER(PaReport)]),
    "pa_jacobi": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64, C.c_double, C.POINTER(PaReport)]),
    "pa_euler_step": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_double, _VP, C.c_double, C.c_double]),
    "pa_euler_march": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_double, _VP, C.c_double, C.c_double, C.c_int64]),
    "pa_cg_begin": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64]),
    "pa_cg_phase_a": (C.c_int, [_VP]),
    "pa_cg_phase_b": (C.c_int, [_VP]),
    "pa_cg_bc": (C.c_int, [_VP]),
    "pa_cg_finish_iter": (C.c_int, [_VP]),
    "pa_cg_iterate": (C.c_int, [_VP, C.c_int64]),
    "pa_cg_fold_plan": (C.c_int, [_VP, _I64P]),
    "pa_cg_fold_set": (C.c_int, [_VP, _I64P]),
    "pa_cg_end": (C.c_int, [_VP, C.POINTER(PaReport)]),
    "pa_slab_set": (C.c_int, [_VP, C.POINTER(PaSlab)]),
    "pa_slab_set_v": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "pa_bicg_begin": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64]),
    "pa_bicg_start": (C.c_int, [_VP]),
    "pa_bicg_pv": (C.c_int, [_VP]),
    "pa_bicg_st": (C.c_int, [_VP]),
    "pa_bicg_x": (C.c_int, [_VP]),
    "pa_bicg_bc": (C.c_int, [_VP]),
    "pa_bicg_finish": (C.c_int, [_VP]),
    "pa_bicg_end": (C.c_int, [_VP, C.POINTER(PaReport)]),
    "pa_comm_available": (C.c_int, []),
    "pa_comm_unique_id": (C.c_int, [_VP]),
    "pa_comm_count": (C.c_int, [_VP, C.POINTER(C.c_int)]),
    "pa_comm_init": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "pa_comm_selftest": (C.c_int, [_VP, C.c_double]),
    "pa_comm_plan": (C.c_int, [_VP, C.POINTER(PaExchange)]),
    "pa_cg_iterate_comm": (C.c_int, [_VP, C.c_int64]),
    "pa_bicg_iterate_comm": (C.c_int, [_VP, C.c_int64]),
    "pa_vec_axpy": (C.c_int, [_VP, _VP, _VP, C.c_double, _VP]),
    "pa_vec_dot": (C.c_int, [_VP, _VP, _VP, C.c_int, C.POINTER(C.c_double)]),
    "pa_vec_mask_interior": (C.c_int, [_VP, _VP]),
    "pa_jacobi_begin": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int64, C.c_double]),
    "pa_jacobi_sweep": (C.c_int, [_VP]),
    "pa_jacobi_bc": (C.c_int, [_VP]),
    "pa_jacobi_finish": (C.c_int, [_VP]),
    "pa_jacobi_end": (C.c_int, [_VP, C.POINTER(PaReport)]),
    "pa_jacobi_iterate_comm": (C.c_int, [_VP, C.c_int64]),
    "pa_comm_destroy": (C.c_int, [_VP]),
    "pa_comm_abort": (C.c_int, [_VP]),
    "pa_stream_wait": (C.c_int, [_VP, C.c_double]),
    "pa_comm_impl": (C.c_char_p, []),
    "pa_comm_use_impl": (C.c_int, [C.c_char_p]),
    "pa_comm_overlap": (C.c_int, [_VP]),
    "pa_report_read": (C.c_int, [_VP, C.POINTER(PaReport)]),
    "pa_scalars_read": (C.c_int, [_VP, _F64P]),
    "pa_profile_set": (C.c_int, [_VP, C.c_int]),
    "pa_profile_read": (C.c_int, [_VP, _F64P, _I64P, _F64P, _I64P]),
    "pa_place_stats": (C.c_int, [_VP, _F64P]),
}

_lib: C.CDLL | None = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libpyapes_hip.so and declare every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PYAPES_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(
            f"pyapes_amd: {p} not found. Build it with pyapes_amd/csrc/build.sh "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
