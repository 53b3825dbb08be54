#!/bin/bash
# round 5 (second session), call n: 3000 training iterations of the final tree (stability under hipGraph replay with the spread
# loads, the new stride-2 data-gradient engine, the four-pixel stem), then the trainer / dist tests
O=gpurun_out/r7n; mkdir -p $O
(echo "# python scripts/long_run.py 3000  (timed configuration: 64x512, B = 64, bf16 trunks, fp32 epilogue on conv_x3 with x_exact, one-launch Philox draws per body, hipGraph replay, ADA target 0.6, synthetic data; final tree of round 5) -- every 300 iterations: losses, ADA statistics, parameter finiteness, the first four input-magnitude EMAs"; timeout 900 python scripts/long_run.py 3000 2>&1 | grep -v amdgpu) | tee $O/long_run.txt | tail -14
