#!/bin/bash
# Copy what a scripts/refresh_profiles.sh call left under gpurun_out/$1 into profiles/round6_* (the files bench.py and the
# docs read).  The listing and the instance JSON carry the stamp of the traced process; nothing is re-hashed here.
set -eu
S=gpurun_out/$1
P=profiles/round6
for f in step_listing.txt step_instances.json pmc.json bench_kernel_stats.csv microbench_tables.txt one_rank_rccl_listing.txt; do
  cp $S/$f ${P}_$f
done
cp $S/prof_stamp.json ${P}_prof_stamp.json
for b in default gfwd 128x1024_bf16 128x1024_fp8 noextra one_rank_rccl; do
  grep '^{' $S/bench_$b.log | tail -1 > ${P}_bench_$b.json
done
ls -la ${P}_*
