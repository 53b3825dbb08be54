#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -10 $O/pytest_gpu.txt
python -c "
import sys; sys.path.insert(0,'.')
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
