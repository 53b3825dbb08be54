#!/bin/bash
# round 6, call o: the suites not yet run on the final structure (ops, full, baselines, metrics, pl, reproducibility, the
# float64 128x1024 tests), then the refresh
O=gpurun_out/r8o; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full.py tests/test_gpu_baselines.py tests/test_gpu_metrics.py tests/test_gpu_pl.py tests/test_gpu_reproducibility.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 1500 python -m pytest tests/test_gpu_128x1024.py -x -q -m gpu -k "float64_oracle or fp32_g_and_d" > $O/tests128.txt 2>&1; echo "tests128 rc=$?"; tail -3 $O/tests128.txt
