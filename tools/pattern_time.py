import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import metafem_jl_amd as mf
for N in (256, 512):
    b = mf.make_Brick((1.0, 1.0, 1.0), (N, N, N))
    torch.cuda.synchronize(); t0 = time.time()
    A = b.pattern(1)
    torch.cuda.synchronize(); print(N, "pattern(1) seconds", round(time.time() - t0, 4), flush=True)
    del A, b
b = mf.make_Brick((1.0, 1.0, 1.0), (128, 128, 128), 2, 5)
torch.cuda.synchronize(); t0 = time.time()
A = b.pattern(1)
torch.cuda.synchronize(); print("hex27 128 pattern(1) seconds", round(time.time() - t0, 4))
