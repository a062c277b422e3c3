#!/bin/bash
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out
for k in 0 4194304 0 4194304; do
echo "== sell knob $k"
MFEM_SELL_KNOB=$k timeout -k 10 300 python tools/u20_leg.py 96 3 2 2>/dev/null | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    o=json.loads(l); print({k: round(o[k],3) for k in ('value','ms_per_step','solve_ms_per_step')}, round(o['roofline'].get('avg_launch_ms',0),4), o['roofline'].get('frac'))"
done
