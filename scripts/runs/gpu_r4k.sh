#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $O/test_ops.txt 2>&1; tail -3 $O/test_ops.txt
python -m pytest tests/test_gpu_pl.py -x -q -m gpu > $O/test_pl.txt 2>&1; tail -3 $O/test_pl.txt
python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu > $O/mb_conv_s2.txt; cat $O/mb_conv_s2.txt
DGV2_WS_NO_GO2=1 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu > $O/mb_conv_s2_nogo2.txt; cat $O/mb_conv_s2_nogo2.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
DGV2_WS_NO_GO2=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_nogo2.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench2.log 2>&1
for f in bench bench_nogo2 bench2; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'))"; done
