#!/bin/bash
# round 5 (second session), call o: conv8_s2d four-class mode on four-row tiles against the two-launch modes: test, microbench
O=gpurun_out/r7o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stride2_data_gradient" > $O/test_s2d.txt 2>&1; echo "s2d test rc=$?"; tail -5 $O/test_s2d.txt
for m in 0 1 2; do echo "--- DGV2_S2D8_MODE=$m"; DGV2_S2D8_MODE=$m timeout 300 python scripts/mb_s2d.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_s2d_mode$m.txt; done
