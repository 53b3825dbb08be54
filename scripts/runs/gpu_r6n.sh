#!/bin/bash
# round 5, call n: reductions folded into the optimizer graphs (one rank on RCCL, A/B in one call); 64-channel tiles for the
# four-class stride-2 data gradients of the deep blocks (experiment: correctness + microbench A/B)
O=gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_trainer.py -x -q -m gpu > $O/test_dist.txt 2>&1; echo "dist/trainer tests rc=$?"; tail -3 $O/test_dist.txt
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_noextra.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_noextra.log') if l.startswith('{')][-1]); print('plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
for v in folded unfolded; do
if [ $v = unfolded ]; then export DGV2_NO_FOLDED_REDUCE=1; fi
DGV2_DIST_WORLD1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_one_rank_rccl_$v.log 2> $O/bench_one_rank_rccl_$v.err; python -c "
import json
try:
    d=json.loads([l for l in open('$O/bench_one_rank_rccl_$v.log') if l.startswith('{')][-1]); print('one-rank rccl $v', round(d['value'],1), round(d['ms_per_step'],3), sorted(d['extra'].get('graphs_live')))
except Exception as e: print('rccl ERR', e)"
done
unset DGV2_NO_FOLDED_REDUCE
echo "--- stride-2 convs, 32-channel four-class tiles (shipped)"; timeout 300 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2_to32.txt
echo "--- DGV2_S2D_TO64=64"; DGV2_S2D_TO64=64 timeout 300 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2_to64.txt
DGV2_S2D_TO64=64 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $O/test_conv_to64.txt 2>&1; echo "conv tests with TO64 rc=$?"; tail -3 $O/test_conv_to64.txt
