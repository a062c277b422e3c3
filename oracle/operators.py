"""Element operators and sparse pattern (oracle; test infrastructure only).

Restates
  * _Var_Basic / _Kval_Basic / _Res_Basic      -- src/solver/06_FEM_Kernel.jl:1-13,28-45,65-79
  * assemble_SparseID! / assemble_KIJ!         -- src/solver/03_GlobalAssembly.jl:77-168
  * sort_CUSPARSE_COO! / generate_J_ptr        -- src/misc/04_GPU_Utils.jl:87-118

Index conventions are 0-based.  ``itg_vals`` is ``integral_vals[q, a, s, host]``
(s = 0 value, 1+d derivative along x_d, see geometry.py); ``s`` plays the role of
the reference's ``sd_IDs`` tuple for max_sd_order = 1.

Difference to the reference (documented, result-neutral): the reference keeps matrix
values in hash-slot order and gathers ``K_total[K_val_ids]`` into CSR order at every
solve (02_Preconditioner.jl:35); here slots ARE the row-sorted CSR positions, i.e.
``K_val_ids`` is the identity.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np


def var_basic(itg_vals, s, cpID_shift, el_g_cpIDs, x, host_ids, el_ids):
    """06_FEM_Kernel.jl:1-13 -> target[q, t] = sum_a N^(s)[q,a,host_t] * x[cp[a,el_t] + shift]."""
    N = itg_vals[:, :, s, :][:, :, host_ids]  # [q, a, t]
    xv = x[el_g_cpIDs[:, el_ids] + cpID_shift]  # [a, t]
    return np.einsum("qat,at->qt", N, xv)


def kval_basic(itg_vals, dual_s, base_s, vals, sparse_IDs_by_el, sparse_ID_shift, K_val, host_ids, el_ids):
    """06_FEM_Kernel.jl:28-45: K[slot[a,b,el]+shift] += sum_q N^(ds)[q,a] N^(bs)[q,b] vals[q,t]."""
    Nd = itg_vals[:, :, dual_s, :][:, :, host_ids]
    Nb = itg_vals[:, :, base_s, :][:, :, host_ids]
    Ke = np.einsum("qat,qbt,qt->abt", Nd, Nb, vals)
    np.add.at(K_val, sparse_IDs_by_el[:, :, el_ids] + sparse_ID_shift, Ke)


def res_basic(itg_vals, dual_s, vals, cpID_shift, el_g_cpIDs, residue, host_ids, el_ids):
    """06_FEM_Kernel.jl:65-79: residue[cp[a,el]+shift] += sum_q N^(ds)[q,a] vals[q,t]."""
    Nd = itg_vals[:, :, dual_s, :][:, :, host_ids]
    fe = np.einsum("qat,qt->at", Nd, vals)
    np.add.at(residue, el_g_cpIDs[:, el_ids] + cpID_shift, fe)


@dataclass
class SparsePattern:
    n: int  # basicfield_size
    unitsize: int  # number of unique (cp_i, cp_j) pairs
    blocks: List[Tuple[int, int]]  # sorted (dual_pos, base_pos) list == sparse_mapping keys
    rowptr: np.ndarray  # [n+1] int64 (0-based CSR)
    colidx: np.ndarray  # [nnz] int32
    pair_I: np.ndarray  # [unitsize] cp row of each unique pair (sorted by (I, J))
    pair_J: np.ndarray
    slot_of_pair: np.ndarray  # [nblocks, unitsize] CSR position of block u / pair p
    pair_by_el: np.ndarray  # [a, b, el] pair id (== sparse_IDs_by_el for block 0 before the CSR sort)

    @property
    def nnz(self) -> int:
        return int(self.colidx.shape[0])

    def sparse_ids_by_el(self, block: Tuple[int, int]) -> np.ndarray:
        """CSR slots [a, b, el] for a field block; plays ``sparse_IDs_by_el + u*unitsize``."""
        u = self.blocks.index(block)
        return self.slot_of_pair[u][self.pair_by_el]


def assemble_sparse_id(cp_ids: np.ndarray, ncp: int, blocks: Sequence[Tuple[int, int]]) -> SparsePattern:
    """Unique (cp_i, cp_j) pairs -> per-block COO -> row-sorted CSR.

    03_GlobalAssembly.jl:77-140 builds the unique pair set with a GPU hash table, then
    ``_assemble_KIJ!`` (:156-168) emits one (I + dual_pos*variable_size,
    J + base_pos*variable_size) entry per pair and block, and ``XcoosortByRow`` +
    ``compress_CSR!`` (04_GPU_Utils.jl:87-118) produce the CSR.  Sorting by (row, col)
    makes the result independent of hash order.
    """
    itp, nel = cp_ids.shape
    I = np.repeat(cp_ids[:, None, :], itp, axis=1).reshape(-1)
    J = np.repeat(cp_ids[None, :, :], itp, axis=0).reshape(-1)
    key = I.astype(np.int64) * ncp + J.astype(np.int64)
    ukey, inv = np.unique(key, return_inverse=True)
    pair_I = (ukey // ncp).astype(np.int64)
    pair_J = (ukey % ncp).astype(np.int64)
    unitsize = ukey.size
    pair_by_el = inv.reshape(itp, itp, nel)
    blocks = sorted(blocks)
    nfields = 1 + max(max(b) for b in blocks)
    n = nfields * ncp
    rows = np.concatenate([pair_I + d * ncp for (d, b) in blocks])
    cols = np.concatenate([pair_J + b * ncp for (d, b) in blocks])
    order = np.lexsort((cols, rows))
    slot = np.empty(order.size, dtype=np.int64)
    slot[order] = np.arange(order.size)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return SparsePattern(n=n, unitsize=unitsize, blocks=list(blocks), rowptr=rowptr,
                         colidx=cols[order].astype(np.int32), pair_I=pair_I, pair_J=pair_J,
                         slot_of_pair=slot.reshape(len(blocks), unitsize), pair_by_el=pair_by_el)
