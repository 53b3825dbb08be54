#!/bin/bash
# round 5 (second session), call e: conv8's load spreading -- over how many taps, and inputs before weights? (variant
# libraries differ in conv8.o only); mb_conv8 rows: s2 fwd x4 | s1 fwd + dgrad x5
O=gpurun_out/r7e; mkdir -p $O
for v in shipped r0 shipped r0; do
if [ $v = shipped ]; then unset DGV2_LIB_PATH; else export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_$v.so; fi
echo "--- $v"
timeout 300 python scripts/mb_conv8.py 2>&1 | grep -v amdgpu.ids | grep -m1 "conv8 image:" | tee -a $O/mb_conv8_$v.txt
done
