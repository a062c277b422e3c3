#!/bin/bash
# round 6, third GPU call: deterministic lattice tiles (mode 5)
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_lat8.py tests/test_gpu_remainder.py tests/test_gpu_slab.py tests/test_gpu_elasticity.py tests/test_gpu_multirank.py -x -q > gpurun_out/t3.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/t3.log; exit 1; }
tail -3 gpurun_out/t3.log
for c in c3 ref_idrs8 nitsche_c2; do
timeout -k 10 300 python bench.py --config $c --steps 8 --warmup 2 --live-traffic 0 --cpu-n 0 --full-out gpurun_out/full_$c.json > gpurun_out/line_$c.json 2> gpurun_out/err_$c.log || { echo "bench $c failed"; tail -5 gpurun_out/err_$c.log; exit 1; }
python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.load(open(f'gpurun_out/full_{c}.json'))
print(c, 'value %.4e ms/step %.2f solve %.2f spmv %.4f frac %.3f' % (d['value'], d['ms_per_step'], d['config']['solve_ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']), d['roofline']['kernel_key'], d['config']['initial_res'], d['config']['final_res'])
PY
done
