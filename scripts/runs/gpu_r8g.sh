#!/bin/bash
# round 6, call g: one-launch synthetic loader, D conv weight gradients written into the flat buffer, d_tail unit tests,
# the 128x1024 fixture test; bench + step listing (launch count)
O=gpurun_out/r8g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_128x1024.py -x -q -m gpu -k "reference_fixture" > $O/t128.txt 2>&1; echo "t128 rc=$?"; tail -3 $O/t128.txt; grep "^E  " $O/t128.txt | cut -c1-300 | head -5
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_gpu_full.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --stamp $R/$O/prof_stamp.json > $R/$O/prof.log 2>&1
cd $R
f=$(ls -t $(find $O/prof -name "*kernel_trace.csv") | head -1); python scripts/step_listing.py $f --full --stamp $O/prof_stamp.json > $O/step_listing.txt; head -8 $O/step_listing.txt; rm -rf $O/prof
