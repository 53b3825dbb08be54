#!/bin/bash
# round 5 (second session), call s: small-channel resample with independent loads, ADA kernels: tests + gfwd / step bench
O=gpurun_out/r7s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_full.py -x -q -m gpu -k "resample or model or full or generator or ada" > $O/test.txt 2>&1; echo "tests rc=$?"; tail -3 $O/test.txt
timeout 300 python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/gfwd.log 2>&1; python -c "
import json; d=json.loads([l for l in open('$O/gfwd.log') if l.startswith('{')][-1]); print('gfwd', round(d['value'],1), round(d['ms_per_step'],4))"
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
