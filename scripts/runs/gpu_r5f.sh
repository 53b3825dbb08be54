#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
run() { n=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/$n.log 2>&1; python -c "
import json; d=json.loads([l for l in open('$O/$n.log') if l.startswith('{')][-1]); print('$n', round(d['value'],1), round(d['ms_per_step'],3))"; }
run base A=1
run tpb1 DGV2_CONV8_TPB=1
run tpb2 DGV2_CONV8_TPB=2
run tpb4 DGV2_CONV8_TPB=4
run wsbig256 DGV2_WS_BLOCKS_BIG=256
run wsbig768 DGV2_WS_BLOCKS_BIG=768
run wssmall256 DGV2_WS_BLOCKS_SMALL=256
run nt0 DGV2_NT_MIN_MB=0
run nt16 DGV2_NT_MIN_MB=16
run nt256 DGV2_NT_MIN_MB=256
run base2 A=1
