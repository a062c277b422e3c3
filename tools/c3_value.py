"""One number for A/B runs: bench.py --config c3 (or c4) value and ms per step, no side legs.  usage: c3_value.py [c3|c4]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--steps", "8", "--warmup", "2", "--live-traffic", "0", "--cpu-n", "0"],
                   capture_output=True, text=True)
o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print(f"{cfg} value {o['value']:.4e} ms_per_step {o['ms_per_step']:.2f} solve {o['config']['solve_ms_per_step']:.2f} spmv {o['roofline']['avg_launch_ms']:.4f}")
