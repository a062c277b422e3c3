"""Sparse multivariate polynomial algebra (oracle; test infrastructure only).

Restates ``src/misc/03_Polynomial.jl`` of the reference: a polynomial is a list
of (factor, exponent-tuple) terms; ``check_Clear`` drops terms with
``|factor| < 1e-8`` (``03_Polynomial.jl:61-75``) -- kept because it decides which
round-off survives in the shape-function tables.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Sequence, Tuple

_CLEAR_ERR = 1e-8  # 03_Polynomial.jl:64


class Poly:
    __slots__ = ("dim", "terms")

    def __init__(self, dim: int, terms: Dict[Tuple[int, ...], float] | None = None):
        self.dim = dim
        self.terms: Dict[Tuple[int, ...], float] = dict(terms) if terms else {}

    # -- constructors -----------------------------------------------------
    @staticmethod
    def const(dim: int, c: float) -> "Poly":
        return Poly(dim, {(0,) * dim: float(c)})

    @staticmethod
    def var(dim: int, i: int) -> "Poly":
        """x_i (0-based i): ``collect_Basis`` 03_Polynomial.jl:19."""
        e = [0] * dim
        e[i] = 1
        return Poly(dim, {tuple(e): 1.0})

    def copy(self) -> "Poly":
        return Poly(self.dim, self.terms)

    # -- check_Clear (03_Polynomial.jl:61-75) ------------------------------
    def _clear(self) -> "Poly":
        self.terms = {e: c for e, c in self.terms.items() if abs(c) >= _CLEAR_ERR}
        return self

    # -- arithmetic ---------------------------------------------------------
    def __add__(self, other):
        out = self.copy()
        if isinstance(other, Poly):
            for e, c in other.terms.items():
                out.terms[e] = out.terms.get(e, 0.0) + c
            return out._clear()
        if other == 0:
            return out
        z = (0,) * self.dim
        out.terms[z] = out.terms.get(z, 0.0) + float(other)
        return out._clear()

    __radd__ = __add__

    def __neg__(self):
        return Poly(self.dim, {e: -c for e, c in self.terms.items()})

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, other):
        if isinstance(other, Poly):
            out: Dict[Tuple[int, ...], float] = {}
            for e1, c1 in self.terms.items():
                for e2, c2 in other.terms.items():
                    e = tuple(a + b for a, b in zip(e1, e2))
                    out[e] = out.get(e, 0.0) + c1 * c2
            return Poly(self.dim, out)._clear()
        if other == 0:
            return Poly(self.dim)
        return Poly(self.dim, {e: c * other for e, c in self.terms.items()})

    __rmul__ = __mul__

    def __truediv__(self, num: float):
        return self * (1.0 / num)  # 03_Polynomial.jl:89

    # -- calculus / evaluation ---------------------------------------------
    def derivative(self, orders: Sequence[int]) -> "Poly":
        """Mixed derivative of the given per-dimension orders (03_Polynomial.jl:127-141)."""
        out: Dict[Tuple[int, ...], float] = {}
        for e, c in self.terms.items():
            if min(a - b for a, b in zip(e, orders)) < 0:
                continue
            f = c
            for a, b in zip(e, orders):
                f *= math.factorial(a) / math.factorial(a - b)
            ne = tuple(a - b for a, b in zip(e, orders))
            out[ne] = out.get(ne, 0.0) + f
        return Poly(self.dim, out)._clear()

    def __call__(self, pos: Iterable[float]) -> float:
        """03_Polynomial.jl:143-149."""
        pos = tuple(pos)
        s = 0.0
        for e, c in self.terms.items():
            t = 1.0
            for p, k in zip(pos, e):
                t *= p ** k
            s += t * c
        return s


def prod(polys: Iterable[Poly]) -> Poly:
    it = iter(polys)
    out = next(it)
    for p in it:
        out = out * p
    return out
