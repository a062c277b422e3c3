"""Where is the assembled hex-8 thermal matrix not bitwise symmetric?"""
import os, sys
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
faces = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x3F
brick = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
A = brick.pattern(1)
K = brick.assemble_thermal(A, 0.6, 25.0, 293.15, faces).cpu().numpy()
M = sp.csr_matrix((K, A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=(A.n, A.n))
D = (M - M.T).tocoo()
nz = D.data != 0
print("n", A.n, "asymmetric entries", int(nz.sum()), "of", M.nnz, "max", float(np.abs(D.data).max()) if D.nnz else 0.0)
m = N + 1
r, c = D.row[nz], D.col[nz]
def ijk(t): return t // (m * m), (t // m) % m, t % m
ri, rj, rk = ijk(r); ci, cj, ck = ijk(c)
onb = lambda a: (a == 0) | (a == N)
print("rows on the boundary:", int((onb(ri) | onb(rj) | onb(rk)).sum()), " both interior:", int((~(onb(ri) | onb(rj) | onb(rk)) & ~(onb(ci) | onb(cj) | onb(ck))).sum()))
for t in range(min(6, len(r))):
    print((int(ri[t]), int(rj[t]), int(rk[t])), (int(ci[t]), int(cj[t]), int(ck[t])), M[r[t], c[t]], M[c[t], r[t]])
