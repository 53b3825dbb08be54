#!/bin/bash
# round 5 (second session), call c: conv_pipe_kernel's F33 forms with the next stage's loads spread over the tap loop
# (shipped) against the burst (libdgv2_cp0.so, -DDGV2_CP_SPREAD=0): microbenchmarks, conv tests, bench line -- one box
O=gpurun_out/r7c; mkdir -p $O
for v in spread burst; do
if [ $v = burst ]; then export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_cp0.so; fi
echo "--- $v"
DGV2_NO_CONV8=1 timeout 300 python scripts/mb_conv8.py 2>&1 | grep -v amdgpu.ids | grep -m2 "conv_pipe:\|us (TF" | tee $O/mb_conv8_pipe_$v.txt
timeout 300 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2_$v.txt
timeout 300 python scripts/mb_conv.py 2>&1 | grep -v amdgpu.ids | grep "k3s1" | tee $O/mb_conv_$v.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$v.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('$v: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
unset DGV2_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $O/test_conv.txt 2>&1; echo "conv tests rc=$?"; tail -3 $O/test_conv.txt
