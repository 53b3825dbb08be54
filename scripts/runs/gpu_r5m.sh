#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
echo "--- shipped"; timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep "B=128" | cut -c1-200
for e in 1 2 3; do echo "--- exp $e (1: no wait states, 2: no sched barriers, 3: both)"; DGV2_LIB_PATH=$PWD/dusty-gan-v2_amd/lib/libdgv2_exp$e.so timeout 300 python scripts/mb_conv_x3.py 2>&1 | grep "B=128" | cut -c1-200; done
