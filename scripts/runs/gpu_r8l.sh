#!/bin/bash
# round 6, call l: fused data gradient + activation backward with the upstream outputs prefetched a sample ahead
O=gpurun_out/r8l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dgrad_with_upstream" > $O/tests_new.txt 2>&1; echo "new tests rc=$?"; tail -3 $O/tests_new.txt
timeout 300 python scripts/mb_dgrad_actbwd.py 2>&1 | grep -v amdgpu.ids > $O/mb_dgrad_actbwd.txt; cat $O/mb_dgrad_actbwd.txt
for v in "" "DGV2_NO_DGRAD_ACTBWD=1" "" "DGV2_NO_DGRAD_ACTBWD=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench $v', d['value'], d['ms_per_step'])"
done
