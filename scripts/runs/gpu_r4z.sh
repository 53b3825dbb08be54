#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grouped_linear or glin or mapping" > $O/test_glin.txt 2>&1; echo "rc=$?"; tail -3 $O/test_glin.txt
DGV2_GLIN_GRAD=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py -q -m gpu > $O/test_model_glingrad.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/test_model_glingrad.txt | head -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
DGV2_GLIN_GRAD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_glingrad.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench2.log 2>$O/bench.err
for f in bench bench_glingrad bench2; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3))"; done
