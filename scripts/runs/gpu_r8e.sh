#!/bin/bash
# round 6, call e: the 128x1024 reference-fixture test again (R1 input-gradient criterion as at 64x512)
O=gpurun_out/r8e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_128x1024.py -x -q -m gpu -k "reference_fixture" > $O/tests2.txt 2>&1; echo "tests2 rc=$?"; tail -5 $O/tests2.txt; grep "^E " $O/tests2.txt | cut -c1-300 | head
