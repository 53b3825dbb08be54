#!/bin/bash
O=gpurun_out/r5a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py tests/test_gpu_pl.py tests/test_gpu_dist.py tests/test_gpu_baselines.py -q -m gpu -x > $O/test_sub.txt 2>&1; echo "rc=$?"; tail -4 $O/test_sub.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err
DGV2_NO_GLIN_GRAD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_noglingrad.log 2>&1
for f in bench bench_noglingrad; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['extra'].items() if k.startswith('ms_') or k.startswith('value_')})"; done
