#!/bin/bash
# round 5 (second session), call j: stride-2 data gradients on the eight-wave engine (conv8_s2d.hip): tests, microbench
O=gpurun_out/r7j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stride2_data_gradient" > $O/test_s2d.txt 2>&1; echo "s2d test rc=$?"; tail -15 $O/test_s2d.txt
timeout 300 python scripts/mb_s2d.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_s2d.txt
