#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_x3" > $O/test_x3.txt 2>&1; echo "rc=$?"; tail -2 $O/test_x3.txt
timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep -v amdgpu | tee $O/mb_conv_x3.txt
timeout 300 python scripts/mb_x3_ablate.py 2>&1 | grep -v amdgpu | tee $O/mb_x3_ablate.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/scripts/mb_conv_x3.py > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_mb.csv; head -12 $O/kernel_stats_mb.csv | cut -c1-160
