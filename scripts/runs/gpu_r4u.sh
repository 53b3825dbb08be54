#!/bin/bash
O=gpurun_out/r4u; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -22 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err
python -c "
import json,sys; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['extra'].items() if k.startswith('ms_') or k.startswith('value_')}, d['roofline']['frac'], d['roofline'].get('stats_measured_on_this_tree'))"
timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3.txt
