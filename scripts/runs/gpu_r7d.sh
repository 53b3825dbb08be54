#!/bin/bash
# round 5 (second session), call d: conv_wgrad_stream_bf16_kernel (64 x 64 asm-MFMA instances) with the next tile's loads spread
# over the (row, tap) loop (shipped) against the burst (libdgv2_ws0.so, -DDGV2_WS_SPREAD=0) -- one box
O=gpurun_out/r7d; mkdir -p $O
for v in spread burst; do
if [ $v = burst ]; then export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_ws0.so; fi
echo "--- $v"
timeout 300 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2_$v.txt
timeout 300 python scripts/mb_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_$v.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$v.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('$v: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
unset DGV2_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or wgrad" > $O/test_conv.txt 2>&1; echo "conv tests rc=$?"; tail -3 $O/test_conv.txt
