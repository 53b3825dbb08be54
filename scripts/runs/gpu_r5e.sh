#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
for args in "--dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra" "--no-graph --steps 5 --warmup 2 --no-cpu-baseline --no-extra" "--batch-per-gpu 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extra" "--d-epilogue bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra"; do
  n=$(echo $args | tr -c 'a-z0-9' '_' | cut -c1-40)
  timeout 600 python bench.py $args > $O/b_$n.log 2>&1; rc=$?
  python -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/b_$n.log') if l.startswith('{')][-1]); print('$args', '->', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), d['dtype'])
except Exception as e: print('$args', 'FAILED rc=$rc', e); print(open('$O/b_$n.log').read()[-1500:])"
done
