#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
R=$GRAFT_REPO_ROOT
python scripts/mb_conv.py 2>&1 | grep -v amdgpu > $O/mb_conv.txt
DGV2_WS_GO2_S1=1 python scripts/mb_conv.py 2>&1 | grep -v amdgpu > $O/mb_conv_go2s1.txt
paste -d'\n' $O/mb_conv.txt $O/mb_conv_go2s1.txt | grep "k3s1"
DGV2_WS_GO2_S1=1 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_bf16_exact or conv_triple" > $O/test_go2s1.txt 2>&1; tail -2 $O/test_go2s1.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err
DGV2_WS_GO2_S1=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_go2s1.log 2>&1
for f in bench bench_go2s1; do python -c "
import json,sys; d=json.loads([l for l in open('$O/$f.log') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; python scripts/prof_buckets.py $O/kernel_stats.csv > $O/buckets.txt; head -40 $O/buckets.txt
