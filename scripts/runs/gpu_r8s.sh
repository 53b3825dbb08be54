#!/bin/bash
# round 6, call s: the round-5 final tree (cb5deb3, built beside this one in _ab_r5/) against the current tree, same box,
# alternating: bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra
O=$GRAFT_REPO_ROOT/gpurun_out/r8s; mkdir -p $O
for rep in 1 2 3; do
  for t in _ab_r5 .; do
    (cd $GRAFT_REPO_ROOT/$t && timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('tree $t', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')" | tee -a $O/ab_r5_vs_r6.txt)
  done
done
