"""mul! (mfem_spmv_csr) on the hex-27 N^3 matrix: time of 20 launches + check against a float64 torch sparse product on a sample of rows."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
if len(sys.argv) > 2:  # mfem_debug_set_spmv knob (variant << 16: 1 product tile, 3 row blocks, 7 wave tiles of a fixed row count | bit 27 no 2688-entry tile)
    from metafem_jl_amd import _lib
    _lib.lib.mfem_debug_set_spmv(int(sys.argv[2], 0), 0)
b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), 2, 5)
A = b.pattern(1)
K = b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
x = mf.FEM_rand(A.n, 3, 0)
y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
for _ in range(3): mf.mul_(y, A, K, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): mf.mul_(y, A, K, x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
nb = A.nnz * 12 + A.n * 16 + (A.n + 1) * 8
print(f"hex-27 {N}^3: {ms:.3f} ms  {nb / ms / 1e6:.0f} GB/s  frac {nb / ms / 1e6 / 8000:.3f}")
rp = A.rowptr[:200001].cpu(); ci = A.colidx[: int(rp[-1])].cpu().long(); kv = K[: int(rp[-1])].cpu(); xc = x.cpu()
ref = torch.zeros(200000, dtype=torch.float64)
rows = torch.repeat_interleave(torch.arange(200000), (rp[1:] - rp[:-1]).long())
ref.index_add_(0, rows, kv * xc[ci])
print("max rel err (first 200k rows)", float((y[:200000].cpu() - ref).abs().max() / ref.abs().max()))
