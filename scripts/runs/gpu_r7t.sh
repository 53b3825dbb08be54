#!/bin/bash
# round 5 (second session), call t: rocprofv3 --kernel-trace --stats of the bench command on the final tree (the per-kernel
# summary the per-body listing is cut from)
O=$GRAFT_REPO_ROOT/gpurun_out/r7t; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -5 $O/kernel_stats.csv | cut -c1-200
rm -rf $O/prof
