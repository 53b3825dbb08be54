#!/bin/bash
# round 5 (second session), call b: conv8 with the next stage's loads spread over the tap loop (shipped) against the burst
# in front of the loop (libdgv2_sp0.so, -DDGV2_C8_SPREAD=0): microbenchmarks, the conv tests, the bench line -- one box
O=gpurun_out/r7b; mkdir -p $O
echo "--- spread (shipped)"; timeout 300 python scripts/mb_conv8.py 2>&1 | grep -v amdgpu.ids | grep "conv8 image:\|us (TF" | tee $O/mb_conv8_spread.txt
echo "--- burst";  DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_sp0.so timeout 300 python scripts/mb_conv8.py 2>&1 | grep -v amdgpu.ids | grep "conv8 image:" | tee $O/mb_conv8_burst.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $O/test_conv.txt 2>&1; echo "conv tests rc=$?"; tail -3 $O/test_conv.txt
for v in spread burst; do
if [ $v = burst ]; then export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_sp0.so; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$v.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('$v: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
