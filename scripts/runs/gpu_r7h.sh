#!/bin/bash
# round 5 (second session), call h: four-pixel stem forward, 256-thread EMA folds: tests + same-box bench A/B (DGV2_STEM_NO4)
O=gpurun_out/r7h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem or ema or modconv or model" > $O/test_ops.txt 2>&1; echo "ops tests rc=$?"; tail -3 $O/test_ops.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full.py -x -q -m gpu > $O/test_model.txt 2>&1; echo "model tests rc=$?"; tail -3 $O/test_model.txt
for v in stem4 stem1 stem4 stem1; do
if [ $v = stem1 ]; then export DGV2_STEM_NO4=1; else unset DGV2_STEM_NO4; fi
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_$v.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('$v: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
