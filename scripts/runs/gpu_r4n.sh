#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_x3 or weight_bank" -s > $O/test_x3.txt 2>&1; echo "rc=$?"; grep -E "conv_x3|passed|failed|Error|error" $O/test_x3.txt | tail -30
timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
DGV2_NO_CONV_X3=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_nox3.log 2>&1
for f in bench bench_nox3; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3))"; done
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 --deselect tests/test_gpu_128x1024.py::test_fp32_g_and_d_step_match_the_oracle_at_128x1024 --deselect tests/test_gpu_128x1024.py::test_e4m3_branches_against_the_float64_oracle_at_128x1024 > $O/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -30 $O/pytest_gpu.txt
