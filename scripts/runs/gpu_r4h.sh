#!/bin/bash
# one gpurun call: modconv_up tests, model / trainer tests, A/B benches of the commuted levels
O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "modconv_up" > $O/test_up.txt 2>&1; tail -5 $O/test_up.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py -x -q -m gpu > $O/test_model.txt 2>&1; tail -5 $O/test_model.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err
DGV2_UP_COMMUTE_O=32 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_up32.log 2>$O/bench_up32.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench2.log 2>&1
python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/gfwd.log 2>&1
DGV2_UP_COMMUTE_O=32 python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/gfwd_up32.log 2>&1
for f in bench bench_up32 bench2 gfwd gfwd_up32; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,2) for k, v in d.get('extra', {}).items() if k.startswith(('ms_', 'value_'))})"; done
