#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python scripts/prof_torch_ops.py > $O/torch_ops.txt 2>&1; tail -60 $O/torch_ops.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r1 -- python3 $R/scripts/prof_r1.py 1 > $R/$O/prof_r1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r0 -- python3 $R/scripts/prof_r1.py 0 > $R/$O/prof_r0.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gfwd -- python3 $R/bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $R/$O/prof_gfwd.log 2>&1
cd $R
cp $(find $O/prof_r1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_r1.csv; cp $(find $O/prof_r0 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_r0.csv; cp $(find $O/prof_gfwd -name "*kernel_stats.csv" | head -1) $O/kernel_stats_gfwd.csv
python scripts/r1_diff.py $O/kernel_stats_r1.csv $O/kernel_stats_r0.csv 14 2>&1 | head -40 | cut -c1-200 | tee $O/r1_diff.txt
head -40 $O/kernel_stats_gfwd.csv | cut -c1-200
