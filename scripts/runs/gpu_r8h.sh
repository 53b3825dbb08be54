#!/bin/bash
# round 6, call h: gemm_x3 with the three-plane split inside the MFMA loop -- test + microbenchmark; d_tail unit tests
O=gpurun_out/r8h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_x3 or d_tail or linear" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 300 python scripts/mb_linear_x3.py 2>&1 | grep -v amdgpu.ids > $O/mb_linear_x3.txt; cat $O/mb_linear_x3.txt
