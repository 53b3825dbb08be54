#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
(echo "# python scripts/long_run.py 3000  (timed configuration: 64x512, B = 64, bf16 trunks, fp32 epilogue on conv_x3 with x_exact, grouped-Linear gradient passes, hipGraph replay, ADA target 0.6, synthetic data) -- every 300 iterations: losses, ADA statistics, parameter finiteness, the first four input-magnitude EMAs"; timeout 900 python scripts/long_run.py 3000 2>&1 | grep -v amdgpu) | tee $O/long_run.txt | tail -14
