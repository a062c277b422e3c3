"""ctypes front end of oracle/c/liboracle.so + the C-backed thermal pipeline used as
``cpu_baseline`` (oracle; test infrastructure only -- see oracle/c/oracle.c)."""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "c", "liboracle.so")
_lib = None

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} missing: run `make -C oracle/c` (or __graft_entry__.build())")
        L = C.CDLL(_LIB)
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_parallel_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_parallel_zero.argtypes = [C.c_void_p, C.c_int64]
        L.orc_update_basic_elements_3d.argtypes = [C.c_int, C.c_int, C.c_int64, f64p, f64p, f64p, C.c_int64, i64p, f64p, f64p]
        L.orc_var_basic.argtypes = [C.c_int, C.c_int, C.c_int, f64p, C.c_int, C.c_int64, i64p, f64p, f64p, i64p, i64p, C.c_int64]
        L.orc_kval_basic.argtypes = [C.c_int, C.c_int, C.c_int, f64p, C.c_int, C.c_int, f64p, i64p, C.c_int64, f64p, i64p, i64p, C.c_int64]
        L.orc_res_basic.argtypes = [C.c_int, C.c_int, C.c_int, f64p, C.c_int, f64p, C.c_int64, i64p, f64p, i64p, i64p, C.c_int64]
        L.orc_scale_weights.argtypes = [C.c_int, C.c_double, f64p, i64p, C.c_int64, f64p]
        L.orc_pattern.argtypes = [C.c_int, C.c_int64, C.c_int64, i64p, i64p, C.c_void_p, C.c_void_p]
        L.orc_pattern.restype = C.c_int64
        L.orc_spmv.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p, C.c_double, C.c_double]
        L.orc_jacobi_by_diagonal.argtypes = [C.c_int64, i64p, i32p, f64p, f64p]
        L.orc_cg_jacobi.argtypes = [C.c_int64, i64p, i32p, f64p, f64p, f64p, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_cg_jacobi.restype = C.c_int
        _lib = L
    return _lib


def rehome(a: np.ndarray) -> np.ndarray:
    """Copy `a` into freshly allocated memory whose pages are first touched by the OpenMP team (NUMA first-touch)."""
    out = np.empty_like(a)
    lib().orc_parallel_copy(out.ctypes.data, a.ctypes.data, a.nbytes)
    return out


def pzeros(n: int, dtype=np.float64) -> np.ndarray:
    out = np.empty(n, dtype=dtype)
    lib().orc_parallel_zero(out.ctypes.data, out.nbytes)
    return out


def _ref_vals_flat(disc) -> np.ndarray:
    """ref_itp_vals[q, a, s] with s = 0 value, 1..3 first derivatives, column-major flattened."""
    r = disc.ref_itp_vals
    out = np.stack([r[:, :, 0, 0, 0], r[:, :, 1, 0, 0], r[:, :, 0, 1, 0], r[:, :, 0, 0, 1]], axis=2)
    return np.asfortranarray(out).ravel(order="F").copy()


@dataclass
class CThermal:
    """Thermal problem of examples/thermal_conduction/3D_Script.jl:30-31 on a make_Brick hex-8 mesh, driven
    through the C restatement with the reference's call sequence."""
    n: tuple
    x: tuple = (1.0, 1.0, 1.0)
    k: float = 0.6
    h: float = 25.0
    Tenv: float = 293.15
    src: float = 1600.0
    itg_order: int = 3

    def setup(self):
        from . import geometry, mesh as om, reference_element as re_

        L = lib()
        self.disc = re_.initialize_classical_element(3, "CUBE", 1, 1, self.itg_order)
        self.mesh = om.lattice_mesh(self.x, self.n, self.disc)
        self.itg, self.itp = self.disc.itg_func_num, self.disc.itp_func_num
        nel, ncp = self.mesh.nel, self.mesh.ncp
        self.cp_ids = np.ascontiguousarray(self.mesh.cp_ids.T).ravel()  # [a + itp*e]
        self.coords = np.ascontiguousarray(self.mesh.coords.T).ravel()  # SoA x|y|z
        self.cp_ids, self.coords = rehome(self.cp_ids), rehome(self.coords)
        self.ivals = pzeros(self.itg * self.itp * 4 * nel)
        self.w = pzeros(self.itg * nel)
        L.orc_update_basic_elements_3d(self.itg, self.itp, nel, _ref_vals_flat(self.disc), self.disc.itg_weight.copy(),
                                       self.coords, ncp, self.cp_ids, self.ivals, self.w)
        # boundary (surface only): numpy geometry, C operators
        self.facets = om.boundary_facets_structured(self.x, self.n, 3)
        fg = geometry.update_basic_boundary(self.mesh, self.disc, self.facets)
        self.f_itg = self.disc.bdy_itg_func_num
        self.f_ivals = np.asfortranarray(fg.integral_vals).ravel(order="F").copy()  # [q, a, s, f]
        self.f_w = np.asfortranarray(fg.integral_weights).ravel(order="F").copy()
        self.f_host = np.arange(len(self.facets), dtype=np.int64)
        self.f_el = self.facets.element_ID.astype(np.int64).copy()
        # pattern + slot table
        self.rowptr = np.empty(ncp + 1, dtype=np.int64)
        nnz = L.orc_pattern(self.itp, nel, ncp, self.cp_ids, self.rowptr, None, None)
        self.colidx = pzeros(nnz, np.int32)
        self.slots = pzeros(self.itp * self.itp * nel, np.int64)
        L.orc_pattern(self.itp, nel, ncp, self.cp_ids, self.rowptr, self.colidx.ctypes.data, self.slots.ctypes.data)
        self.rowptr = rehome(self.rowptr)
        self.el_ids = rehome(np.arange(nel, dtype=np.int64))
        self.K = pzeros(nnz)
        self.residue = pzeros(ncp)
        self.xstar = pzeros(ncp)
        self.s = rehome(np.full(ncp, self.src))
        self._vals = pzeros(self.itg * nel)
        self._fvals = pzeros(self.f_itg * len(self.facets))
        return self

    def K_linear_func(self):
        L = lib()
        nel = self.mesh.nel
        L.orc_parallel_zero(self.K.ctypes.data, self.K.nbytes)
        for d in range(3):  # three _Kval_Basic launches: (T;d, T;d) with coefficient -k
            L.orc_scale_weights(self.itg, -self.k, self.w, self.el_ids, nel, self._vals)
            L.orc_kval_basic(self.itg, self.itp, 4, self.ivals, 1 + d, 1 + d, self._vals, self.slots, 0, self.K,
                             self.el_ids, self.el_ids, nel)
        nf = len(self.facets)
        L.orc_scale_weights(self.f_itg, -self.h, self.f_w, self.f_host, nf, self._fvals)
        L.orc_kval_basic(self.f_itg, self.itp, 4, self.f_ivals, 0, 0, self._fvals, self.slots, 0, self.K, self.f_host,
                         self.f_el, nf)

    def K_nonlinear_func(self):
        L = lib()
        nel, nf = self.mesh.nel, len(self.facets)
        L.orc_parallel_zero(self.residue.ctypes.data, self.residue.nbytes)
        for d in range(3):  # T;d word, then residual term -k*T;d on dual T;d
            buf = pzeros(self.itg * nel)
            L.orc_var_basic(self.itg, self.itp, 4, self.ivals, 1 + d, 0, self.cp_ids, self.xstar, buf, self.el_ids, self.el_ids, nel)
            buf *= -self.k
            buf *= self.w
            L.orc_res_basic(self.itg, self.itp, 4, self.ivals, 1 + d, buf, 0, self.cp_ids, self.residue, self.el_ids, self.el_ids, nel)
        buf = pzeros(self.itg * nel)  # external s
        L.orc_var_basic(self.itg, self.itp, 4, self.ivals, 0, 0, self.cp_ids, self.s, buf, self.el_ids, self.el_ids, nel)
        buf *= self.w
        L.orc_res_basic(self.itg, self.itp, 4, self.ivals, 0, buf, 0, self.cp_ids, self.residue, self.el_ids, self.el_ids, nel)
        fb = np.zeros(self.f_itg * nf)  # boundary: h*(Tenv - T)
        L.orc_var_basic(self.f_itg, self.itp, 4, self.f_ivals, 0, 0, self.cp_ids, self.xstar, fb, self.f_host, self.f_el, nf)
        fb = self.h * (self.Tenv - fb) * self.f_w
        L.orc_res_basic(self.f_itg, self.itp, 4, self.f_ivals, 0, fb, 0, self.cp_ids, self.residue, self.f_host, self.f_el, nf)

    def solve_cg(self, tol: float, maxiter: int, fixed: bool = False):
        L = lib()
        x = pzeros(self.mesh.ncp)
        res = C.c_double()
        it = L.orc_cg_jacobi(self.mesh.ncp, self.rowptr, self.colidx, self.K, self.residue, x, tol, maxiter, 1 if fixed else 0,
                             C.byref(res))
        return x, it, res.value

    def timed_step(self, iters: int):
        """One benchmark step (assembly + fixed CG iterations); returns (t_assembly, t_solve)."""
        t0 = time.perf_counter()
        self.K_linear_func()
        self.K_nonlinear_func()
        t1 = time.perf_counter()
        self.solve_cg(0.0, iters, fixed=True)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1
