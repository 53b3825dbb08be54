#!/bin/bash
# round 6, call f: fused D tail (tests of the model / trainer / full suites), R1 input gradient at 128x1024 vs the oracle
O=gpurun_out/r8f; mkdir -p $O
timeout 600 python scripts/dbg/r1_128.py > $O/r1_128.txt 2>&1; tail -8 $O/r1_128.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_gpu_full.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'])"
DGV2_NO_D_TAIL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_off.log 2>$O/bench_off.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_off.log') if l.startswith('{')][-1]); print('bench (tail off)', d['value'], d['ms_per_step'])"
