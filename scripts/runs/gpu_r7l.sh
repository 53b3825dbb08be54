#!/bin/bash
# round 5 (second session), call l: conv_wgrad_x3_kernel with the next tile's loads spread over the (row, tap) loop (shipped)
# against the burst (libdgv2_wx0.so): microbench, x3 tests, bench A/B
O=gpurun_out/r7l; mkdir -p $O
for v in spread burst spread burst; do
if [ $v = burst ]; then export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_wx0.so; else unset DGV2_LIB_PATH; fi
echo "--- $v"; timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu.ids | tee -a $O/mb_conv_x3_$v.txt
done
unset DGV2_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "x3" > $O/test_x3.txt 2>&1; echo "x3 tests rc=$?"; tail -3 $O/test_x3.txt
for v in spread burst spread burst; do
if [ $v = burst ]; then export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_wx0.so; else unset DGV2_LIB_PATH; fi
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_$v.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('$v: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
