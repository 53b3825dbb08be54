#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grouped_linear or style_affines" > $O/test_glin.txt 2>&1; tail -15 $O/test_glin.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py tests/test_gpu_pl.py -x -q -m gpu > $O/test_model.txt 2>&1; tail -5 $O/test_model.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
DGV2_NO_GLIN=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_noglin.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench2.log 2>&1
python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/gfwd.log 2>&1
DGV2_NO_GLIN=1 python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/gfwd_noglin.log 2>&1
for f in bench bench_noglin bench2 gfwd gfwd_noglin; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3))"; done
