#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
for i in 1 2 3; do
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_128x1024.py::test_fp32_g_and_d_step_match_the_oracle_at_128x1024 --deselect tests/test_gpu_128x1024.py::test_e4m3_branches_against_the_float64_oracle_at_128x1024 -p no:randomly > $O/pytest_$i.txt 2>&1; echo "run $i rc=$?"; tail -2 $O/pytest_$i.txt
done
