#!/bin/bash
# round 6, call i: gemm_x3 forms per shape (microbench), the bf16-vs-float64 gradient table, the whole ops / model /
# trainer / full suites, bench + listing
O=gpurun_out/r8i; mkdir -p $O
timeout 300 python scripts/mb_linear_x3.py 2>&1 | grep -v amdgpu.ids > $O/mb_linear_x3.txt; cat $O/mb_linear_x3.txt
timeout 600 python scripts/bf16_grad_table.py > $O/bf16_vs_f64_table.txt 2> $O/bf16_table.err; grep "^==\|bf16 mode" $O/bf16_vs_f64_table.txt; tail -3 $O/bf16_table.err
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_gpu_full.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --stamp $R/$O/prof_stamp.json > $R/$O/prof.log 2>&1
cd $R
f=$(ls -t $(find $O/prof -name "*kernel_trace.csv") | head -1); python scripts/step_listing.py $f --full --stamp $O/prof_stamp.json > $O/step_listing.txt; head -8 $O/step_listing.txt; rm -rf $O/prof
