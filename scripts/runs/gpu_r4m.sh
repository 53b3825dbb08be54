#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu --durations=25 --deselect tests/test_gpu_128x1024.py::test_fp32_g_and_d_step_match_the_oracle_at_128x1024 > $O/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -45 $O/pytest_gpu.txt
timeout 900 python -m pytest tests/test_gpu_128x1024.py -x -q -m gpu -k "fp32_g_and_d" > $O/pytest_128_fp32.txt 2>&1; echo "rc=$?"; tail -15 $O/pytest_128_fp32.txt
