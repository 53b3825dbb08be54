#!/bin/bash
# round 5 (second session), call f: per-body listing of the step after the load spreading
O=$GRAFT_REPO_ROOT/gpurun_out/r7f; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_listing.py $f --full --json $O/step_instances.json > $O/step_listing.txt; head -40 $O/step_listing.txt
rm -rf $O/prof
