#!/bin/bash
# round 6, call p: runtime launch-path knobs on the replayed step (kernel arguments in device memory, AQL packet capture
# of graphs, hardware queue count) -- same box, same tree
O=gpurun_out/r8p; mkdir -p $O
for v in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8" "X=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench $v', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $O/knobs.txt
done
