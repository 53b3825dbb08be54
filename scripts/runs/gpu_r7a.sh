#!/bin/bash
# round 5 (second session), call a: where conv8's stride-2 forward instances spend their time today (ablation build on the
# weight image), tiles per block, and the box's baseline lines
O=gpurun_out/r7a; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_noextra.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_noextra.log') if l.startswith('{')][-1]); print('plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 300 python scripts/mb_s2_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/s2_probe_tpb.txt
timeout 600 python scripts/mb_s2_probe.py dusty-gan-v2_amd/lib/libdgv2_abl.so 2>&1 | grep -v amdgpu.ids | tee $O/s2_probe_abl.txt
timeout 300 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2.txt
timeout 300 python scripts/mb_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv.txt
