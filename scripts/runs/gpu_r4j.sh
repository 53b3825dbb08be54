#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grouped_linear or style_affines or conv" > $O/test_ops.txt 2>&1; tail -4 $O/test_ops.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py tests/test_gpu_pl.py -x -q -m gpu > $O/test_model.txt 2>&1; tail -4 $O/test_model.txt
python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu > $O/mb_conv_s2.txt; cat $O/mb_conv_s2.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/gfwd.log 2>&1
for f in bench gfwd; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'))"; done
