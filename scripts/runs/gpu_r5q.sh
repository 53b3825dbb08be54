#!/bin/bash
O=gpurun_out/r5q; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_x3" > $O/test_x3.txt 2>&1; echo "rc=$?"; tail -2 $O/test_x3.txt
bash scripts/refresh_profiles.sh r5q > $O/refresh.log 2>&1
for f in bench_default bench_gfwd bench_128x1024_bf16 bench_128x1024_fp8 bench_one_rank_rccl; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('kernel','')[:40])
except Exception as e: print('$f', 'ERR', e)"; done
head -30 $O/buckets.txt; cat $O/pmc.json | head -40
