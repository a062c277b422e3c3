"""On-box probe: can a process that has initialised the GPU start GPU-using child processes (subprocess = fork + exec in the
child), and can two processes share cuda:0?"""
import subprocess, sys, time
import torch
print("parent cuda:", torch.cuda.is_available(), torch.cuda.get_device_name(0))
x = torch.ones(4, device="cuda"); print("parent sum", float(x.sum()))
t0 = time.time()
child = "import torch,os;x=torch.ones(1000,device='cuda');print('child',os.getpid(),float(x.sum()))"
ps = [subprocess.Popen([sys.executable, "-c", child], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for _ in range(2)]
for p in ps:
    out, _ = p.communicate(timeout=300)
    print("rc", p.returncode, out.decode()[-400:])
print("elapsed", time.time() - t0)
