#!/bin/bash
# round 5 (second session), call g: the stride-2 data gradient (conv_pipe, four classes) with its loads spread over the taps
O=gpurun_out/r7g; mkdir -p $O
for v in spread4 burst spread4 burst; do
if [ $v = burst ]; then export DGV2_LIB_PATH=dusty-gan-v2_amd/lib/libdgv2_s40.so; else unset DGV2_LIB_PATH; fi
echo "--- $v"
timeout 300 python scripts/mb_conv_s2.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2_$v.txt
timeout 300 python scripts/mb_conv_s2.py 64 2>&1 | grep -v amdgpu.ids | tee $O/mb_conv_s2_b64_$v.txt
done
unset DGV2_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > $O/test_conv.txt 2>&1; echo "conv tests rc=$?"; tail -3 $O/test_conv.txt
