#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_full.py -x -q -m gpu -k "r1_pass_with_the_epilogue" -s > $O/t.txt 2>&1; echo "rc=$?"; tail -15 $O/t.txt
