"""ms per launch of the CSR kernel behind mul! on configs[2] (hex-8 elasticity 128^3) and configs[3] (hex-27 128^3): what tools/rb_sweep.sh times per library variant."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
lam, mu = 0.5769230769230769, 0.38461538461538464
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, order, F in (("c3", 1, 3), ("c4", 2, 1)):
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N), order, 3 if order == 1 else 5)
    A = b.pattern(F)
    K = b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS["x0"]) if F == 3 else b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F)
    x = mf.FEM_rand(A.n, 3, 0) - 0.5
    y = torch.zeros(A.n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        mf.mul_(y, A, K, x)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            mf.mul_(y, A, K, x)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    byts, _ = A.spmv_bytes()
    print(f"matrix ms {name} {best:.4f}  ({byts / best / 1e6 / 8000:.3f} of 8 TB/s on {byts / 1e9:.2f} GB)  checksum {float(y.double().sum()):.10e}", flush=True)
    del b, A, K, x, y
    torch.cuda.empty_cache()
