"""hex-8 assembly kernels (thermal K + R at 256^3 and 512^3, elasticity K + R at 128^3), hip-event times per call incl. faces.  usage: hex8_asm_time.py [256|512|c3 ...]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
which = sys.argv[1:] or ["256", "c3"]
for w in which:
    if w == "c3":
        lam, mu = 0.5769230769230769, 0.38461538461538464
        b = mf.make_Brick((1.0, 1.0, 1.0), (128, 128, 128), 1, 3)
        A = b.pattern(3)
        K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
        x = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        print(f"elasticity 128^3: K {t(lambda: b.assemble_elasticity(A, lam, mu, 1000.0, mf.FACE_BITS['x0'], out=K)):.3f} ms  "
              f"R {t(lambda: b.residual_elasticity(x, lam, mu, 1000.0, mf.FACE_BITS['x0'], mf.FACE_BITS['y1'], (0.0, 1.0, 0.0, 0.0, 0.0, 0.0))):.3f} ms", flush=True)
    else:
        N = int(w)
        b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
        A = b.pattern(1)
        K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
        x = torch.zeros(A.n, dtype=torch.float64, device="cuda")
        s = torch.full((A.n,), 1600.0, dtype=torch.float64, device="cuda")
        print(f"thermal {N}^3: K {t(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)):.3f} ms  R {t(lambda: b.residual_thermal(x, 0.6, 25.0, 293.15, 0x3F, s=s)):.3f} ms", flush=True)
    del b, A, K
    torch.cuda.empty_cache()
