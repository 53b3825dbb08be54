#!/bin/bash
# round 5, call m: distributed tests after the tail fix, one-rank RCCL A/B inside one call
O=gpurun_out/r6m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_trainer.py -x -q -m gpu > $O/test_dist.txt 2>&1; echo "dist/trainer tests rc=$?"; tail -3 $O/test_dist.txt
for i in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_noextra_$i.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_noextra_$i.log') if l.startswith('{')][-1]); print('plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
DGV2_DIST_WORLD1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_one_rank_rccl_$i.log 2> $O/bench_one_rank_rccl_$i.err; python -c "
import json
try:
    d=json.loads([l for l in open('$O/bench_one_rank_rccl_$i.log') if l.startswith('{')][-1]); print('one-rank rccl', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('graphs_live'))
except Exception as e: print('rccl ERR', e)"
done
