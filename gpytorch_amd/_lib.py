"""ctypes binding of ``libgpamd.so`` (the C ABI declared in ``include/gpamd.h``).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
The product path never routes through ``oracle/`` or any CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPAMD_LIBRARY: A/B builds of the same ABI (tuning sessions only); the product is csrc/libgpamd.so
LIB_PATH = os.environ.get("GPAMD_LIBRARY") or os.path.join(_HERE, "csrc", "libgpamd.so")

_lib = None

_i = C.c_int
_i64 = C.c_int64
_p = C.c_void_p
_f = C.c_float

# name -> (restype, argtypes); mirrors include/gpamd.h one-to-one
SIGNATURES = {
    "gpamd_abi_version": (_i, []),
    "gpamd_last_error": (C.c_char_p, []),
    "gpamd_prep_points_f32": (_i, [_i, _f, _p, _i, _i, _i64, _p, _i, _p, _p, _i, _p]),
    "gpamd_kv_plan": (_i, [_i, _i, _i, _i, _i, _i, _i64, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64)]),
    "gpamd_kv_partials_f32": (_i, [_i, _f, _p, _i, _p, _i, _i, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _p, _p]),
    "gpamd_kv_partials_far_f32": (_i, [_i, _f, _p, _i, _p, _i, _i, _p, _p, _i64, _i, _p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _i64]),
    "gpamd_kv_far_workspace_ints": (_i64, [_i, _i, _i]),
    "gpamd_kv_reduce_f32": (_i, [_p, _i, _i64, _i, _i, _p, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    "gpamd_kv_f32": (_i, [_i, _f, _p, _i, _p, _i, _i, _p, _p, _i64, _i, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i, _p]),
    "gpamd_kernel_rows_f32": (_i, [_i, _f, _p, _p, _i, _p, _i, _i, _p, _p, _i64, _p]),
    "gpamd_kernel_dense_f32": (_i, [_i, _f, _p, _i, _p, _i, _i, _p, _p, _i64, _p]),
    "gpamd_kernel_diag_f32": (_i, [_i, _f, _p, _p, _i, _i, _p, _p, _p]),
    "gpamd_coldot_f32": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    "gpamd_cg_fscratch_elems": (_i64, [_i, _i]),
    "gpamd_cg_iscratch_elems": (_i64, [_i]),
    "gpamd_cg_layout": (_i, [_i, _i, C.POINTER(_i64)]),
    "gpamd_cg_create_f32": (_p, [_i, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _i, _f, _f]),
    "gpamd_cg_destroy": (None, [_p]),
    "gpamd_cg_done_ptr": (_p, [_p]),
    "gpamd_cg_init_f32": (_i, [_p, _p, _i64, _i, _p]),
    "gpamd_cg_partials_layout": (_i, [_i, _i, _i, _p, _p, _p]),
    "gpamd_cg_init_norms_f32": (_i, [_p, _p, _i64, _p]),
    "gpamd_cg_init_apply_f32": (_i, [_p, _p, _i64, _i, _p]),
    "gpamd_cg_begin_apply_f32": (_i, [_p, _p]),
    "gpamd_cg_dot_rz_f32": (_i, [_p, _p]),
    "gpamd_cg_update_d_apply_f32": (_i, [_p, _i, _p]),
    "gpamd_cg_begin_f32": (_i, [_p, _p]),
    "gpamd_cg_reduce_q_f32": (_i, [_p, _p, _i, _i64, _p, _p, _p, _p]),
    "gpamd_cg_update_xr_f32": (_i, [_p, _i, _p]),
    "gpamd_cg_update_d_f32": (_i, [_p, _i, _p]),
    "gpamd_cg_stop_f32": (_i, [_p, _i, _i, _i, _f, _p]),
    "gpamd_cg_stop_comm_f32": (_i, [_p, _i, _i, _i, _f, _p, _p]),
    "gpamd_allreduce_sum_f32": (_i, [_p, _i64, _p, _p]),
    "gpamd_cg_finish_f32": (_i, [_p, _p]),
    "gpamd_pivoted_cholesky_f32": (_i, [_i, _f, _p, _i, _i, _p, _i, _f, _p, _i64, _p, _p, _p, _p]),
    "gpamd_lanczos_num_partials": (_i, [_i]),
    "gpamd_lanczos_partial_stride": (_i, []),
    "gpamd_lanczos_residual_f32": (_i, [_p, _p, _p, _p, _i, _p]),
    "gpamd_lanczos_project_f32": (_i, [_p, _i64, _i, _p, _i, _p, _p]),
    "gpamd_lanczos_coef_f32": (_i, [_p, _i, _i, _f, _p, _p, _p]),
    "gpamd_lanczos_subtract_f32": (_i, [_p, _i64, _i, _p, _p, _i, _p, _p]),
    "gpamd_lanczos_normalize_f32": (_i, [_p, _i, _p, _p, _p, _f, _p, _p]),
    "gpamd_block_project_f32": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _p, _p, _i64, _p]),
    "gpamd_block_project_f64": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _p, _p, _i64, _p]),
    "gpamd_block_subtract_f32": (_i, [_p, _i64, _i, _p, _p, _i64, _i, _i, _p]),
    "gpamd_block_transform_f32": (_i, [_p, _p, _i64, _i, _i, _p]),
    "gpamd_msminres_update_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _p]),
    "gpamd_precond_coef_workspace_doubles": (_i64, [_i, _i, _i]),
    "gpamd_precond_coef_f32f64": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _p, _p, _i64, _p]),
    "gpamd_precond_apply_f32f64": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _p, _p, _p, _i64, _p]),
    "gpamd_kv_grad_workspace_doubles": (_i64, [_i, _i, _i, _i]),
    # ---- float64 section
    "gpamd_kernel_dense_batched_f32": (_i, [_i, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _i64, _p]),
    "gpamd_kernel_grad_batched_f32": (_i, [_i, _p, _p, _i, _p, _i, _i, _i, _p, _i64, _p, _p]),
    "gpamd_prep_points_f64": (_i, [_i, C.c_double, _p, _i, _i, _i64, _p, _i, _p, _p, _i, _p]),
    "gpamd_kernel_rows_f64": (_i, [_i, C.c_double, _p, _p, _i64, _i, _p, _i, _i, _p, _p, _i64, _p]),
    "gpamd_kernel_diag_f64": (_i, [_i, C.c_double, _p, _p, _i, _i, _p, _p, _p]),
    "gpamd_kv_plan_f64": (_i, [_i, _i, _i, _i, _i64, _p, _p, _p]),
    "gpamd_kv_partials_f64": (_i, [_i, C.c_double, _p, _i, _p, _i, _i, _p, _i64, _i, _p, _i64, _i, _i, _p, _p]),
    "gpamd_kernel_grad_block_f32": (_i, [_i, C.c_double, _p, _i64, _i, _p, _i, _i, _p, _i64, _p, _p]),
    "gpamd_kernel_grad_block_f64": (_i, [_i, C.c_double, _p, _i64, _i, _p, _i, _i, _p, _i64, _p, _p]),
    "gpamd_coldot_f64": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    "gpamd_kv_reduce_f64": (_i, [_p, _i, _i64, _i, _i, _p, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    "gpamd_cg64_fscratch_elems": (_i64, [_i, _i]),
    "gpamd_cg64_create": (_p, [_i, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _i, C.c_double, C.c_double]),
    "gpamd_cg64_destroy": (None, [_p]),
    "gpamd_cg64_init": (_i, [_p, _p, _i64, _i, _p]),
    "gpamd_cg64_begin": (_i, [_p, _p]),
    "gpamd_cg64_reduce_q": (_i, [_p, _p, _i, _i64, _p, _p, _p, _p]),
    "gpamd_cg64_update_xr": (_i, [_p, _i, _p]),
    "gpamd_cg64_update_d": (_i, [_p, _i, _p]),
    "gpamd_cg64_stop": (_i, [_p, _i, _i, _i, C.c_double, _p]),
    "gpamd_cg64_finish": (_i, [_p, _p]),
    "gpamd_kv_grad_f32": (_i, [_i, _p, _i, _p, _i, _i, _p, _i64, _p, _i64, _i, _i, _p, _p, _i64, _p]),
    "gpamd_kv_grad2_workspace_doubles": (_i64, [_i, _i, _i, _i]),
    "gpamd_kv_grad2_xworkspace_floats": (_i64, [_i, _i, _i, _i]),
    "gpamd_kv_grad2_f32": (_i, [_i, _f, _p, _i, _p, _i, _i, _p, _p, _i64, _p, _i64, _i, _i, _p, _p, _i64, _p, _i64, _p, _i64, _i, _p, _i64, _p]),
    "gpamd_kv_grad2_split_workspace_floats": (_i64, [_i, _i]),
    "gpamd_kv_grad2_far_workspace_ints": (_i64, [_i, _i]),
    "gpamd_kv_grad2_far_f32": (_i, [_i, _f, _p, _i, _p, _i, _i, _p, _p, _i64, _p, _i64, _i, _i, _p, _p, _i64, _p, _i64, _p, _i64, _i, _p, _i64, _p,
                                    _p, _p, _p, _p, _f, _p, _i64]),
    "gpamd_kv_grad_far_workspace_ints": (_i64, [_i, _i]),
    "gpamd_kv_grad_far_f32": (_i, [_i, _p, _i, _p, _i, _i, _p, _i64, _p, _i64, _i, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _f, _p, _i64]),
}


class GpamdError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GpamdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gpytorch_amd/csrc -j8`). gpytorch_amd has no CPU fallback."
            )
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if h.gpamd_abi_version() != 5:
            raise GpamdError("libgpamd.so ABI version mismatch")
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().gpamd_last_error().decode("utf-8", "replace")
        raise GpamdError(f"libgpamd call failed ({what}) rc={rc}: {msg}")
