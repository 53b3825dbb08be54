#!/bin/bash
# round 5 (second session), call q: split count of the streaming weight gradient (DGV2_WS_BLOCKS_BIG: 512 shipped): fewer splits
# = less partial-sum traffic (75 MB per launch at 512 blocks) but one block per CU
O=gpurun_out/r7q; mkdir -p $O
for n in 512 384 256; do
echo "--- DGV2_WS_BLOCKS_BIG=$n"
DGV2_WS_BLOCKS_BIG=$n timeout 300 python scripts/mb_conv.py 2>&1 | grep -v amdgpu.ids | grep "k3s1" | tee $O/mb_conv_$n.txt
DGV2_WS_BLOCKS_BIG=$n timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > $O/bench_$n.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_$n.log') if l.startswith('{')][-1]); print('$n: plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
done
