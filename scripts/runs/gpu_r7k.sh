#!/bin/bash
# round 5 (second session), call k: conv8_s2d with the occupancy rule: test, microbench, same-box bench A/B (DGV2_NO_S2D8)
O=gpurun_out/r7k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stride2_data_gradient or conv_bf16_exact" > $O/test_s2d.txt 2>&1; echo "s2d test rc=$?"; tail -5 $O/test_s2d.txt
timeout 300 python scripts/mb_s2d.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_s2d.txt
for v in s2d8 four s2d8 four; do
if [ $v = four ]; then export DGV2_NO_S2D8=1; else unset DGV2_NO_S2D8; fi
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_$v.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('$v: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
