#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_x3" -s > $O/test_x3.txt 2>&1; echo "rc=$?"; grep -E "wgrad|passed|failed|Error|assert" $O/test_x3.txt | tail -12
timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err
DGV2_NO_WGRAD_X3=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_nowx3.log 2>&1
for f in bench bench_nowx3; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['extra'].items() if k.startswith('ms_') or k.startswith('value_')})"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r1 -- python3 $R/scripts/prof_r1.py 1 > $R/$O/prof_r1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r0 -- python3 $R/scripts/prof_r1.py 0 > $R/$O/prof_r0.log 2>&1
cd $R
cp $(find $O/prof_r1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_r1.csv; cp $(find $O/prof_r0 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_r0.csv
tail -1 $O/prof_r1.log; tail -1 $O/prof_r0.log
python scripts/r1_diff.py $O/kernel_stats_r1.csv $O/kernel_stats_r0.csv 14 2>&1 | head -50 | tee $O/r1_diff.txt
