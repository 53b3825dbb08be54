#!/bin/bash
# round 5, call j: experiment -- the weight bank in the R1 pass (DGV2_R1_BANK=1): which tests survive, what it buys
O=gpurun_out/r6j; mkdir -p $O
DGV2_R1_BANK=1 timeout 600 python -m pytest tests/test_gpu_full.py tests/test_gpu_trainer.py -x -q -m gpu -k "r1 or R1 or trainer or step" > $O/test_r1bank.txt 2>&1; echo "R1-bank tests rc=$?"; tail -25 $O/test_r1bank.txt
DGV2_R1_BANK=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_r1bank.log 2> $O/bench_r1bank.err; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r6j/bench_r1bank.log') if l.startswith('{')][-1])
    print('bench with R1 bank', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('ms_plain_iteration'), d['extra'].get('ms_r1_iteration'))
except Exception as e: print('bench ERR', e)
PY
tail -5 $O/bench_r1bank.err
