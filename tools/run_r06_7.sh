#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out

python - <<'PY'
import json, subprocess, sys
for knob in ("", str(1 << 21), str(2 << 21), "8", ""):
    out = subprocess.run([sys.executable, "tools/u20_leg.py", "96", "3", "2"], capture_output=True, text=True, env=dict(__import__("os").environ, MFEM_SELL_KNOB=knob)).stdout
    for ln in out.splitlines():
        if ln.startswith("{"):
            o = json.loads(ln)
            print("sell knob", knob or "default", {k: round(o[k], 3) if isinstance(o[k], float) else o[k] for k in ("value", "ms_per_step", "solve_ms_per_step")}, o["roofline"]["kernel_key"], round(o["roofline"]["avg_launch_ms"], 3), round(o["roofline"]["frac"], 3), "design GB", round(o["roofline"]["algorithmic_bytes_per_launch"] / 1e9, 2))
PY
