#!/bin/bash
# round 6, call u: how often does the discriminator fall into the D = 0 state on the synthetic scans in the first 600
# iterations -- round-5 final tree (_ab_r5) against the current tree, five seeds each
O=$GRAFT_REPO_ROOT/gpurun_out/r8u; mkdir -p $O
for seed in 0 1 2 3 4; do
  for t in _ab_r5 .; do
    echo "== tree $t seed $seed" >> $O/seeds.txt
    (cd $GRAFT_REPO_ROOT/$t && DGV2_SEED=$seed timeout 300 python scripts/long_run.py 600 2>&1 | grep "^[0-9]" | sed -n '5p;10p' | cut -c1-230 >> $O/seeds.txt)
  done
done
cat $O/seeds.txt
