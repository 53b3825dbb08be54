#!/bin/bash
# round 6, call x: the driver's multi-rank command line on the one-GPU box (two ranks on cuda:0 over gloo: DGV2_DIST_SMOKE) --
# a functional rehearsal of `torch.distributed.run ... bench.py --gpus N`, never a measurement; then one rank on RCCL
O=gpurun_out/r8x; mkdir -p $O
DGV2_DIST_SMOKE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 8 --warmup 2 --batch-per-gpu 16 > $O/smoke2.log 2> $O/smoke2.err; echo "rc=$?"; grep "^{" $O/smoke2.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('2 ranks (gloo, one GPU):', round(d['value'], 1), d['n_gpus'], d['config']['parallelism'], d['extra'].get('graphs_live'), d['extra'].get('backend'))"
tail -2 $O/smoke2.err
DGV2_DIST_WORLD1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/rccl1.log 2> $O/rccl1.err; echo "rc=$?"; grep "^{" $O/rccl1.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('1 rank on RCCL:', round(d['value'], 1), d['extra'].get('graphs_live'), d['extra'].get('captured_collectives'))"
