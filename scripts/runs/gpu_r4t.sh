#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_x3" > $O/test_x3.txt 2>&1; echo "rc=$?"; tail -3 $O/test_x3.txt
echo "--- wave-specialised"; timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3_ws.txt
echo "--- every wave stages"; DGV2_X3_NO_WS=1 timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3_nows.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
DGV2_X3_NO_WS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_nows.log 2>&1
for f in bench bench_nows; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), (d.get('roofline') or {}).get('frac'))"; done
