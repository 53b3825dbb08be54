#!/bin/bash
# round 5, call o: 3000 iterations of the timed configuration (stability with the one-launch RNG, the folded paths), rng
# microbench after the grid cap, a DIST long run on one rank
O=gpurun_out/r6o; mkdir -p $O
timeout 120 python scripts/mb_rng.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_rng.txt
(echo "# python scripts/long_run.py 3000  (timed configuration: 64x512, B = 64, bf16 trunks, fp32 epilogue on conv_x3 with x_exact, one-launch Philox draws per body, hipGraph replay, ADA target 0.6, synthetic data) -- every 300 iterations: losses, ADA statistics, parameter finiteness, the first four input-magnitude EMAs"; timeout 900 python scripts/long_run.py 3000 2>&1 | grep -v amdgpu) | tee $O/long_run.txt | tail -14
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rng" 2>&1 | tail -2
