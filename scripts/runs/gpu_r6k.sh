#!/bin/bash
# round 5, call k: full GPU suite (R1 on the weight bank, the D step on the G step's bank, the x_exact test at B = 64, the
# 128x1024 step at B = 4) + smoke
O=gpurun_out/r6k; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/test_all.txt 2>&1; echo "gpu suite rc=$?"; tail -16 $O/test_all.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r6k/bench_default.log') if l.startswith('{')][-1])
    print('bench', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('ms_plain_iteration'), d['extra'].get('ms_r1_iteration'))
except Exception as e: print('bench ERR', e)
PY
tail -3 $O/bench_default.err
