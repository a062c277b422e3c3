"""Slab domain decomposition host logic (new; the reference is single-GPU).

A structured lattice of m0 x m1 x m2 control points is cut along i into `world` contiguous runs of node
planes.  Rank r owns planes [lo, hi): its matrix rows are its owned nodes (field-major), assembled from
the element planes touching them (the interface element plane is evaluated on both sides, so assembly
needs no communication).  A local vector is laid out as

    [ field 0 owned | field 1 owned | ... | f0 ghost_lo | f0 ghost_hi | f1 ghost_lo | f1 ghost_hi | ... ]

with `order` ghost planes (order * m1*m2 entries: a row couples to control points up to `order` planes away) per
side and field -- a low block holding planes lo-order .. lo-1 and a high block holding hi .. hi+order-1; the ghost
block is always allocated, edge ranks simply never reference their outer half.  Slabs of an order-2 lattice start
and end on element boundaries (even planes).  `slab_local_index` is the host restatement of the device
numbering (csrc/brick.h::brick_xindex) and is what the CPU (gloo) tests exercise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def slab_planes(m0: int, world: int, rank: int, order: int = 1) -> Tuple[int, int]:
    """Owned node planes [lo, hi) of `rank`.  order 1: m0 planes split as evenly as possible, extras to the low ranks.
    order p: the (m0 - 1) / p element planes are split that way and a slab is the planes [p * e_lo, p * e_hi) -- it starts
    and ends on an element boundary; the last rank also owns the closing plane m0 - 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if order > 1:
        if (m0 - 1) % order:
            raise ValueError("m0 is not order * n_elements + 1")
        ne = (m0 - 1) // order
        if ne < world:
            raise ValueError("fewer element planes than ranks")
        base, rem = divmod(ne, world)
        elo = rank * base + min(rank, rem)
        ehi = elo + base + (1 if rank < rem else 0)
        return order * elo, (m0 if rank == world - 1 else order * ehi)
    if m0 < world:
        raise ValueError("fewer node planes than ranks")
    base, rem = divmod(m0, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def slab_local_index(i, j, k, f, lo: int, hi: int, m1: int, m2: int, n_fields: int, order: int = 1):
    """Local vector index of field f at lattice node (i, j, k) for the slab [lo, hi); i in [lo-order, hi+order)."""
    i, j, k = np.asarray(i), np.asarray(j), np.asarray(k)
    pl = m1 * m2
    n_owned = (hi - lo) * pl
    inplane = j * m2 + k
    owned = (i >= lo) & (i < hi)
    side = np.where(i < lo, 0, 1)
    off = np.where(i < lo, i - (lo - order), i - hi)
    return np.where(owned, f * n_owned + (i - lo) * pl + inplane,
                    n_fields * n_owned + ((f * 2 + side) * order + off) * pl + inplane)


def local_vector_length(lo: int, hi: int, m1: int, m2: int, n_fields: int, order: int = 1) -> int:
    return n_fields * ((hi - lo) + 2 * order) * m1 * m2


class SlabComm:
    """RCCL communicator of the C ABI (mfem_comm_*) bootstrapped through torch.distributed: rank 0 creates the
    128-byte unique id, the default process group broadcasts it."""

    def __init__(self, ctx, brick, rank: int, world: int, n_fields: int = 1):
        import torch.distributed as dist

        self.ctx, self.rank, self.world = ctx, rank, world
        dev = f"cuda:{ctx.device}"
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            check(lib.mfem_comm_unique_id(buf))
            uid.copy_(torch.tensor(list(buf), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, src=0)
        raw = bytes(uid.cpu().tolist())
        self._h = C.c_void_p()
        check(lib.mfem_comm_create(ctx._h, rank, world, raw, C.byref(self._h)))
        # the halo is `itp_order` control-point planes thick: the exchange moves that many first / last owned planes per field
        self.plane_len = brick.itp_order * brick.m[1] * brick.m[2]
        self.n_owned_nodes = brick.n_owned
        self.n_fields = n_fields
        check(lib.mfem_context_set_comm(ctx._h, self._h, self.n_owned_nodes, self.plane_len, n_fields))
        ctx._children.add(self)

    def allreduce_(self, t: torch.Tensor) -> torch.Tensor:
        check(lib.mfem_allreduce_sum(self.ctx._h, t.data_ptr(), t.numel()))
        return t

    def halo_(self, x_local: torch.Tensor) -> torch.Tensor:
        check(lib.mfem_halo_exchange(self.ctx._h, x_local.data_ptr()))
        return x_local

    def halo_reduce_(self, x_local: torch.Tensor) -> torch.Tensor:
        check(lib.mfem_halo_reduce(self.ctx._h, x_local.data_ptr()))
        return x_local

    def close(self):
        if self._h:
            lib.mfem_context_set_comm(self.ctx._h, None, 0, 0, 0)
            lib.mfem_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostSlabComm:
    """The same communicator behind host callbacks (mfem_comm_create_host): the library stages device data through pinned
    host memory and torch.distributed (any backend that moves CPU tensors: gloo) does the exchange.  Every solver code path
    is the RCCL one; only the transport differs.  This is how the multi-rank path runs with several ranks on ONE GPU
    (RCCL rejects duplicate devices) -- tests/test_gpu_multirank.py -- and what a host without GPU-aware MPI would use.
    poison = True: ghost entries are NaN between halo begin and end (a kernel that reads them too early is found out)."""

    def __init__(self, ctx, brick, rank: int, world: int, n_fields: int = 1, group=None, poison: bool = False):
        import numpy as _np
        import torch.distributed as dist

        self.ctx, self.rank, self.world, self.group = ctx, rank, world, group
        self.calls = {"allreduce": 0, "exchange": 0}

        def _tensor(ptr, count):
            return torch.from_numpy(_np.ctypeslib.as_array(ptr, shape=(count,)))

        def _allreduce(_user, buf, count):
            try:
                self.calls["allreduce"] += 1
                dist.all_reduce(_tensor(buf, count), group=self.group)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self.error = e
                return 1

        def _exchange(_user, s_lo, r_lo, s_hi, r_hi, count):
            try:
                self.calls["exchange"] += 1
                reqs = []
                if r_lo:
                    reqs.append(dist.irecv(_tensor(r_lo, count), self.rank - 1, group=self.group))
                if r_hi:
                    reqs.append(dist.irecv(_tensor(r_hi, count), self.rank + 1, group=self.group))
                if s_lo:
                    reqs.append(dist.isend(_tensor(s_lo, count), self.rank - 1, group=self.group))
                if s_hi:
                    reqs.append(dist.isend(_tensor(s_hi, count), self.rank + 1, group=self.group))
                for q in reqs:
                    q.wait()
                return 0
            except Exception as e:
                self.error = e
                return 1

        self.error = None
        self._cb = (_lib.ALLREDUCE_CB(_allreduce), _lib.EXCHANGE_CB(_exchange))  # keep the thunks alive
        ops = _lib.CommHostOps(None, self._cb[0], self._cb[1], _lib.COMM_HOST_POISON_GHOSTS if poison else 0, 0)
        self._h = C.c_void_p()
        check(lib.mfem_comm_create_host(ctx._h, rank, world, C.byref(ops), C.byref(self._h)))
        self.plane_len = brick.itp_order * brick.m[1] * brick.m[2]
        self.n_owned_nodes = brick.n_owned
        self.n_fields = n_fields
        check(lib.mfem_context_set_comm(ctx._h, self._h, self.n_owned_nodes, self.plane_len, n_fields))
        ctx._children.add(self)

    allreduce_ = SlabComm.allreduce_
    halo_ = SlabComm.halo_

    def halo_reduce_(self, x_local: torch.Tensor) -> torch.Tensor:
        check(lib.mfem_halo_reduce(self.ctx._h, x_local.data_ptr()))
        return x_local

    close = SlabComm.close
    __del__ = SlabComm.__del__
