#!/bin/bash
# round 5, call l: the tail exchange as a captured body in distributed mode (one rank on RCCL), then the profile refresh
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_trainer.py -x -q -m gpu > $O/test_dist.txt 2>&1; echo "dist/trainer tests rc=$?"; tail -3 $O/test_dist.txt
bash scripts/refresh_profiles.sh r6l > $O/refresh.log 2>&1; cat $O/refresh.log | grep -v "us x" | head -60
