#!/bin/bash
# round 5 (second session), call i: the whole GPU suite on the current tree
O=gpurun_out/r7i; mkdir -p $O
timeout 1100 python -m pytest tests -x -q -m gpu > $O/test_gpu.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/test_gpu.txt
