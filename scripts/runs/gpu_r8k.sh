#!/bin/bash
# round 6, call k: microbenchmark of the fused data gradient + activation backward; step listing with it on
O=gpurun_out/r8k; mkdir -p $O
timeout 300 python scripts/mb_dgrad_actbwd.py 2>&1 | grep -v amdgpu.ids > $O/mb_dgrad_actbwd.txt; cat $O/mb_dgrad_actbwd.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --stamp $R/$O/prof_stamp.json > $R/$O/prof.log 2>&1
cd $R
f=$(ls -t $(find $O/prof -name "*kernel_trace.csv") | head -1); python scripts/step_listing.py $f --full --stamp $O/prof_stamp.json > $O/step_listing.txt; head -8 $O/step_listing.txt; rm -rf $O/prof
