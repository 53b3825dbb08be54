#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_x3 or weight_bank" -s > $O/test_x3.txt 2>&1; echo "rc=$?"; grep -E "conv_x3|passed|failed|Error|error|assert" $O/test_x3.txt | tail -30
timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err
DGV2_NO_WGRAD_X3=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_nowx3.log 2>&1
for f in bench bench_nowx3; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['extra'].items() if k.startswith('ms_') or k.startswith('value_')})"; done
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_full.py tests/test_gpu_model.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest_sub.txt 2>&1; echo "rc=$?"; tail -5 $O/pytest_sub.txt
