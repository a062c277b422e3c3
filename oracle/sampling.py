"""Point sampling of a finite-element field (oracle; test infrastructure only).

The reference's example scripts compare with line samples taken in Paraview ("plot over line" on the written VTK: quadratic
tetrahedra / hexahedra, i.e. the element's own shape functions) and commit them as CSV files -- cylinder_flow/MetaFEM_y2.csv,
thermal_conduction/MetaFEM_a.csv, stress_concentration/3D_MetaFEM_x.csv, ... (e.g. cylinder_flow/3D_MetaFEM_Script.jl:122-123).
This module evaluates u_h(x) = sum_a N_a(xi(x)) u_a at arbitrary points: candidate elements from a k-d tree of the element
centroids, the isoparametric map inverted by Newton with the discretization's own basis polynomials
(spatial_discretization/102_Interpolations.jl semantics, via reference_element / simplex)."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
from scipy.spatial import cKDTree


class Sampler:
    def __init__(self, mesh, disc, k: int = 24):
        self.mesh, self.disc, self.k = mesh, disc, k
        self.dim = disc.dim
        X = mesh.coords[mesh.cp_ids]  # [itp, nel, dim]
        self.cent = X.mean(axis=0)
        self.tree = cKDTree(self.cent)
        self.funcs = disc.itp_funcs
        self.dfuncs = [[f.derivative(tuple(1 if j == d else 0 for j in range(self.dim))) for d in range(self.dim)] for f in self.funcs]

    def _basis(self, xi) -> Tuple[np.ndarray, np.ndarray]:
        N = np.array([f(xi) for f in self.funcs])
        dN = np.array([[df(xi) for df in row] for row in self.dfuncs])  # [itp, dim]
        return N, dN

    def _inside(self, xi, tol) -> bool:
        if self.disc.shape == "SIMPLEX":
            return bool(np.all(xi >= -tol) and xi.sum() <= 1.0 + tol)
        return bool(np.all(xi >= -tol) and np.all(xi <= 1.0 + tol))

    def locate(self, x: np.ndarray, tol: float = 1e-6):
        """(element, xi) of the element that contains x, or (-1, None)."""
        _, cand = self.tree.query(x, k=min(self.k, self.mesh.nel))
        xi0 = np.full(self.dim, 0.25 if self.disc.shape == "SIMPLEX" else 0.5)
        for e in np.atleast_1d(cand):
            Xe = self.mesh.coords[self.mesh.cp_ids[:, e]]  # [itp, dim]
            xi = xi0.copy()
            ok = False
            for _ in range(25):
                N, dN = self._basis(xi)
                r = N @ Xe - x
                J = Xe.T @ dN  # d x_i / d xi_m
                try:
                    step = np.linalg.solve(J, r)
                except np.linalg.LinAlgError:
                    break
                xi = xi - step
                if np.abs(step).max() < 1e-13:
                    ok = True
                    break
                if np.abs(xi).max() > 5.0:
                    break
            if ok and self._inside(xi, tol):
                return int(e), xi
        return -1, None

    def sample(self, fields: Dict[str, np.ndarray], points: np.ndarray, tol: float = 1e-6):
        """fields: name -> nodal values [ncp]; points [np, dim].  Returns (dict name -> values [np], valid mask)."""
        npnt = points.shape[0]
        out = {k: np.full(npnt, np.nan) for k in fields}
        valid = np.zeros(npnt, dtype=bool)
        for i in range(npnt):
            e, xi = self.locate(points[i], tol)
            if e < 0:
                continue
            N, _ = self._basis(xi)
            ids = self.mesh.cp_ids[:, e]
            for k, v in fields.items():
                out[k][i] = N @ v[ids]
            valid[i] = True
        return out, valid
