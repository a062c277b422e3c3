"""Assembly kernels of the three big configs, hip-event times (A/B of library variants: tools/ab_libs.sh tools/asm_time.py <dir>)."""
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
which = sys.argv[1] if len(sys.argv) > 1 else "c2c4"
if "c2" in which:
    b = mf.make_Brick((1.0, 1.0, 1.0), (256, 256, 256))
    A = b.pattern(1)
    K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
    print("matrix ms 1 c2 hex-8 thermal 256^3", t(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K)))
    del b, A, K
    torch.cuda.empty_cache()
if "c4" in which:
    b = mf.make_Brick((1.0, 1.0, 1.0), (128, 128, 128), 2, 5)
    A = b.pattern(1)
    K = torch.empty(A.nnz, dtype=torch.float64, device="cuda")
    print("matrix ms 2 c4 hex-27 thermal 128^3", t(lambda: b.assemble_thermal(A, 0.6, 25.0, 293.15, 0x3F, out=K), 3))
