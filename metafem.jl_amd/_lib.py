"""ctypes binding of libmetafem_mi355x.so (include/metafem_mi355x.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises.
torch is imported first so that the process-wide libamdhip64 / librccl are torch's copies
(same SONAME as /opt/rocm's; the library binds to whichever is already loaded).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime with the caller's tensors)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmetafem_mi355x.so")


class MetaFEMError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C metafem.jl_amd/csrc`).  There is no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


class _Lib:
    """The loaded library.  The 21 tuning knobs are ONE C entry point since round 6 -- mfem_debug_set(key, a, b), include/metafem_mi355x_debug.h --;
    `lib.mfem_debug_set_<key>(a[, b])` stays available here as a spelling of it (tests, tools and bench.py were written against one function per knob)."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        if name.startswith("mfem_debug_set_"):
            key = name[len("mfem_debug_set_"):]
            cdll = self._cdll

            def knob(a=0, b=0, _key=key):
                if _key == "recheck_scale":
                    return cdll.mfem_debug_set(b"recheck_scale_ppm", int(round(float(a) * 1e6)), 0)
                return cdll.mfem_debug_set(_key.encode(), int(a), int(b))
            return knob
        return getattr(self._cdll, name)


lib = _Lib(_load())

c_void_p, c_int, c_int32, c_int64, c_double, c_uint64, c_uint32 = (
    C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_double, C.c_uint64, C.c_uint32)


class SolveOptions(C.Structure):
    _fields_ = [("method", c_int32), ("precond", c_int32), ("l_or_s", c_int32), ("maxiter", c_int32),
                ("max_pass", c_int32), ("check_every", c_int32), ("converge_tol", c_double), ("seed", c_uint64),
                ("fixed_iterations", c_int32), ("scale_in_place", c_int32), ("left_precond", c_int32),
                ("cg_variant", c_int32)]


class SolveStats(C.Structure):
    _fields_ = [("passes", c_int32), ("iterations", c_int32), ("final_res", c_double), ("initial_res", c_double),
                ("solve_ms", c_double), ("converged", c_int32), ("spmv_count", c_int32)]


class ThermalParams(C.Structure):
    _fields_ = [("k", c_double), ("h", c_double), ("Tenv", c_double), ("robin_faces", c_uint32), ("fixed_faces", c_uint32),
                ("h_penalty", c_double), ("Tw", c_double)]


class ElasticityParams(C.Structure):
    _fields_ = [("lam", c_double), ("mu", c_double), ("tau", c_double), ("penalty_faces", c_uint32),
                ("traction_faces", c_uint32), ("sig", c_double * 6)]


class KvalTerm(C.Structure):
    _fields_ = [("dual_sd", c_int32), ("base_sd", c_int32), ("block", c_int32), ("reserved", c_int32)]


class ResBatchTerm(C.Structure):
    _fields_ = [("dual_sd", c_int32), ("reserved", c_int32), ("cpID_shift", c_int64)]


class VarBatchTerm(C.Structure):
    _fields_ = [("sd", c_int32), ("reserved", c_int32), ("cpID_shift", c_int64), ("x", C.c_void_p)]


class ConstTerm(C.Structure):
    _fields_ = [("dual_sd", c_int32), ("base_sd", c_int32), ("block", c_int32), ("reserved", c_int32), ("coef", c_double)]


MAX_BATCH_TERMS = 48


ALLREDUCE_CB = C.CFUNCTYPE(c_int, c_void_p, C.POINTER(c_double), c_int32)
EXCHANGE_CB = C.CFUNCTYPE(c_int, c_void_p, C.POINTER(c_double), C.POINTER(c_double), C.POINTER(c_double), C.POINTER(c_double), c_int64)


class CommHostOps(C.Structure):
    _fields_ = [("user", c_void_p), ("allreduce_sum", ALLREDUCE_CB), ("neighbour_exchange", EXCHANGE_CB), ("flags", c_uint32),
                ("reserved", c_uint32)]


COMM_HOST_POISON_GHOSTS = 1


class OpLayout(C.Structure):
    _fields_ = [("itg", c_int32), ("itp", c_int32), ("n_sd", c_int32), ("n_host", c_int64), ("index_base", c_int32),
                ("n_colours", c_int32), ("colour_offsets", C.POINTER(c_int64))]


# every symbol declared in include/metafem_mi355x.h and include/metafem_mi355x_debug.h: (name, restype, argtypes)
P = c_void_p
SIGNATURES = {
    "mfem_abi_version": (c_int, []),
    "mfem_last_error": (C.c_char_p, []),
    "mfem_context_create": (c_int, [c_int, P, C.POINTER(P)]),
    "mfem_context_set_stream": (c_int, [P, P]),
    "mfem_context_destroy": (c_int, [P]),
    "mfem_context_sync": (c_int, [P]),
    "mfem_csr_create": (c_int, [P, c_int64, c_int64, P, c_int, P, c_int, C.POINTER(P)]),
    "mfem_csr_destroy": (c_int, [P]),
    "mfem_spmv_csr": (c_int, [P, P, P, P, P, c_double, c_double]),
    "mfem_axpby": (c_int, [P, c_int64, c_double, P, c_double, P]),
    "mfem_dot": (c_int, [P, c_int64, P, P, C.POINTER(c_double)]),
    "mfem_nrm2": (c_int, [P, c_int64, P, C.POINTER(c_double)]),
    "mfem_rand": (c_int, [P, c_int64, c_uint64, c_uint32, P]),
    "mfem_jacobi_by_diagonal": (c_int, [P, P, P, P]),
    "mfem_jacobi2_by_column": (c_int, [P, P, P, P]),
    "mfem_jacobi_by_row": (c_int, [P, P, P, P]),
    "mfem_mat_div_jacobi": (c_int, [P, P, P, P]),
    "mfem_debug_hex27_direct_count": (c_int64, []),
    "mfem_debug_hex27_mixed_count": (c_int64, []),
    "mfem_debug_hex27_rows_count": (c_int64, []),
    "mfem_debug_mesh_rows_count": (c_int64, []),
    "mfem_debug_bsell_fields": (c_int, [P]),
    "mfem_debug_bsell_spmv_count": (c_int64, []),
    "mfem_debug_sell_periodic_blocks": (c_int64, [P]),
    "mfem_debug_symp_fingerprint_count": (C.c_longlong, []),
    "mfem_debug_set": (c_int, [C.c_char_p, c_int64, c_int64]),
    "mfem_debug_graph_comm_count": (c_int, []),
    "mfem_debug_lat27_cg_fused": (c_int, []),
    "mfem_debug_lat27_pass1_bytes": (c_int64, [P]),
    "mfem_prof_spmv_enable": (c_int, [P, c_int]),
    "mfem_prof_spmv_read": (c_int, [P, C.POINTER(c_double), C.POINTER(c_int64), c_int]),
    "mfem_prof_comm_enable": (c_int, [P, c_int]),
    "mfem_prof_comm_read": (c_int, [P, C.POINTER(c_double), C.POINTER(c_int64), C.POINTER(c_double), C.POINTER(c_int64), c_int]),
    "mfem_solve": (c_int, [P, P, P, P, P, C.POINTER(SolveOptions), C.POINTER(SolveStats)]),
    "mfem_solve_set_shadow": (c_int, [P, P, c_int32]),
    "mfem_brick_create": (c_int, [P, c_int32, c_int32, c_int32, c_double, c_double, c_double, c_int32, c_int32,
                                  C.POINTER(P)]),
    "mfem_brick_destroy": (c_int, [P]),
    "mfem_brick_num_controlpoints": (c_int64, [P]),
    "mfem_brick_num_elements": (c_int64, [P]),
    "mfem_brick_coords": (P, [P, c_int32]),
    "mfem_brick_set_slab": (c_int, [P, c_int32, c_int32]),
    "mfem_brick_pattern": (c_int, [P, P, c_int32, C.POINTER(P)]),
    "mfem_pattern_build": (c_int, [P, c_int32, c_int64, c_int64, P, c_int32, c_int32, C.POINTER(P), P]),
    "mfem_csr_rowptr64": (P, [P]),
    "mfem_csr_colidx": (P, [P]),
    "mfem_csr_nnz": (c_int64, [P]),
    "mfem_csr_n": (c_int64, [P]),
    "mfem_brick_assemble_thermal": (c_int, [P, P, P, C.POINTER(ThermalParams), P]),
    "mfem_brick_residual_thermal": (c_int, [P, P, C.POINTER(ThermalParams), P, P, P]),
    "mfem_brick_assemble_elasticity": (c_int, [P, P, P, C.POINTER(ElasticityParams), P]),
    "mfem_brick_residual_elasticity": (c_int, [P, P, C.POINTER(ElasticityParams), P, P]),
    "mfem_update_basic_elements": (c_int, [P, c_int32, c_int32, c_int32, c_int64, c_int64, P, P, P, P, c_int32, P, P]),
    "mfem_update_basic_boundary": (c_int, [P, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, P, P, P, P, P, P, P,
                                          c_int32, P, P, P]),
    "mfem_op_var": (c_int, [P, C.POINTER(OpLayout), P, c_int32, c_int64, P, P, P, P, P, c_int64]),
    "mfem_op_kval": (c_int, [P, C.POINTER(OpLayout), P, c_int32, c_int32, P, P, c_int64, P, P, P, c_int64]),
    "mfem_op_res": (c_int, [P, C.POINTER(OpLayout), P, c_int32, P, c_int64, P, P, P, P, c_int64]),
    "mfem_debug_lat27_spmv_count": (c_int64, []),
    "mfem_debug_lat27_asymmetry": (C.c_double, [c_void_p]),
    "mfem_debug_lat8_spmv_count": (c_int64, []),
    "mfem_debug_lat8_asymmetry": (C.c_double, [c_void_p]),
    "mfem_spmv_solver_layout": (c_int, [P, P, P, P, P, c_double, c_double]),
    "mfem_csr_solver_layout": (c_int, [P, P, C.POINTER(c_int32), C.POINTER(c_int32), C.POINTER(c_int64), C.POINTER(c_int64)]),
    "mfem_csr_solver_layout_entries": (c_int, [P, P, C.POINTER(c_int64), C.POINTER(c_int32)]),
    "mfem_csr_solver_layout_bytes": (c_int, [P, P, C.POINTER(c_int64)]),
    "mfem_debug_sym_spmv_count": (c_int64, []),
    "mfem_op_kval_batch": (c_int, [P, C.POINTER(OpLayout), P, c_int32, C.POINTER(KvalTerm), P, P, c_int64, c_int64, P, P, P, c_int64]),
    "mfem_op_res_batch": (c_int, [P, C.POINTER(OpLayout), P, c_int32, C.POINTER(ResBatchTerm), P, P, P, P, P, c_int64]),
    "mfem_op_var_batch": (c_int, [P, C.POINTER(OpLayout), P, c_int32, C.POINTER(VarBatchTerm), P, P, P, P, c_int64]),
    "mfem_mesh_assemble_elements": (c_int, [P, c_int32, c_int32, c_int32, c_int64, c_int64, P, P, P, P, c_int32, c_int32,
                                            C.POINTER(ConstTerm), P, c_int64, P, P, c_int64, c_int32, C.POINTER(c_int64)]),
    "mfem_mesh_assemble_elements_rows": (c_int, [P, c_int32, c_int32, c_int32, c_int64, c_int64, P, P, P, P, c_int32, c_int32,
                                                 C.POINTER(ConstTerm), c_int32, P, P, P, P, P]),
    "mfem_mesh_assemble_elements_rows_set": (c_int, [P, c_int32, c_int32, c_int32, c_int64, c_int64, P, P, P, P, c_int32, c_int32,
                                                 C.POINTER(ConstTerm), c_int32, P, P, P, P, P]),
    "mfem_mesh_row_ranks": (c_int, [P, c_int32, c_int64, c_int64, c_int32, P, P, P, P, c_int32, P]),
    "mfem_mesh_assemble_facets": (c_int, [P, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, P, P, P, P, P, P, P, c_int32,
                                          c_int32, C.POINTER(ConstTerm), P, c_int64, P, P, c_int64, c_int32, C.POINTER(c_int64)]),
    "mfem_comm_unique_id": (c_int, [P]),
    "mfem_comm_create": (c_int, [P, c_int32, c_int32, P, C.POINTER(P)]),
    "mfem_comm_destroy": (c_int, [P]),
    "mfem_context_set_comm": (c_int, [P, P, c_int64, c_int64, c_int32]),
    "mfem_allreduce_sum": (c_int, [P, P, c_int32]),
    "mfem_halo_exchange": (c_int, [P, P]),
    "mfem_halo_reduce": (c_int, [P, P]),
    "mfem_comm_create_host": (c_int, [P, c_int32, c_int32, C.POINTER(CommHostOps), C.POINTER(P)]),
    "mfem_csr_ncols": (c_int64, [P]),
    "mfem_csr_replan": (c_int, [P, P]),
    "mfem_csr_spmv_bytes": (c_int, [P, P, C.POINTER(c_int64), C.POINTER(c_int64)]),
    "mfem_debug_comm_selftest": (c_int, [P, c_int64, c_int32]),
    "mfem_debug_ws_address": (C.c_ulonglong, [P]),
    "mfem_debug_fail_host_alloc": (c_int, [c_int]),
    "mfem_debug_ws_trial_log": (c_int, [P, C.POINTER(C.c_double)]),
    "mfem_debug_remainder_info": (c_int, [P, C.POINTER(c_int64), C.POINTER(c_int64), C.POINTER(C.c_double)]),
    "mfem_debug_rem_spmv_count": (C.c_longlong, []),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc: int) -> None:
    if rc != 0:
        raise MetaFEMError(f"libmetafem_mi355x rc={rc}: {lib.mfem_last_error().decode()}")
