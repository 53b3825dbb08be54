#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reproducibility.py -x -q -m gpu -k "conv_x3" > $O/test_x3.txt 2>&1; echo "rc=$?"; tail -3 $O/test_x3.txt
timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err
python -c "
import json,sys; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['extra'].items() if k.startswith('ms_') or k.startswith('value_')})"
