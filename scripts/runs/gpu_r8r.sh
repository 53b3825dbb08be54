#!/bin/bash
# round 6, call r: 3 000 training iterations under hipGraph replay on the final structure (fused step graphs, fused kernels of
# this round): losses, ADA statistics, parameter finiteness, status word
O=gpurun_out/r8r; mkdir -p $O
timeout 900 python scripts/long_run.py 3000 2>&1 | grep -v amdgpu.ids > $O/long_run_3000.txt; tail -4 $O/long_run_3000.txt
