#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grouped_linear or glin or mapping" > $O/test_glin.txt 2>&1; echo "rc=$?"; tail -3 $O/test_glin.txt
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py -x -q -m gpu > $O/test_model.txt 2>&1; echo "rc=$?"; tail -3 $O/test_model.txt
python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/bench_gfwd.log 2>&1
python -c "
import json,sys; d=json.loads([l for l in open('$O/bench_gfwd.log') if l.startswith('{')][-1]); print('gfwd', round(d['value'],1), round(d['ms_per_step'],4))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
python -c "
import json,sys; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],3))"
